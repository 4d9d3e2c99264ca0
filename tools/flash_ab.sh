#!/bin/bash
# flash_pipe A/B on one box: the two-waves kernel, the pipelined kernel (default, two blocks per wave, tracking pass) and the
# in-kernel clock / cycles per tile of the stamped builds (tools/flash_variants.sh, tools/flash_stamps.py)
cd "$(dirname "$0")/.."
S="${1:-32 5 9216}"
echo "== two-waves kernel"; DC_FLASH_PIPE=0 python tools/flash_bench.py $S 2>&1 | grep flash
echo "== pipe (default)"; python tools/flash_bench.py $S 2>&1 | grep flash
echo "== pipe, two blocks per wave"; DC_FLASH_QB2=1 python tools/flash_bench.py $S 2>&1 | grep flash
echo "== pipe, tracking pass thr=8"; DC_FLASH_TRACK=1 python tools/flash_bench.py $S 2>&1 | grep flash
for v in tools/_variants/libdc_st*.so; do [ -e $v ] || continue; for q in 0 1; do echo -n "$(basename $v) QB2=$q: "; DC_FLASH_QB2=$q DC_HIP_LIB=$PWD/$v python tools/flash_stamps.py $S 2>&1 | grep flash_pipe; done; done
