#!/bin/bash
# flash_pipe A/B on one box: the two-waves kernel, the pipelined kernel (default and tracking pass), the timing-only variants
# and the in-kernel clock / cycles per tile of the stamped builds (tools/flash_variants.sh, tools/flash_stamps.py)
cd "$(dirname "$0")/.."
S="${1:-32 5 9216}"
echo "== two-waves kernel"; DC_FLASH_PIPE=0 python tools/flash_bench.py $S 2>&1 | grep flash
echo "== pipe (default: first-half-tile shift)"; python tools/flash_bench.py $S 2>&1 | grep flash
echo "== pipe, tracking pass thr=8"; DC_FLASH_TRACK=1 python tools/flash_bench.py $S 2>&1 | grep flash
for v in tools/_variants/libdc_st*.so; do [ -e $v ] || continue; echo -n "$(basename $v): "; DC_HIP_LIB=$PWD/$v python tools/flash_stamps.py $S 2>&1 | grep flash_pipe; done
