#!/bin/bash
# Tool builds of the one-wave-per-SIMD GEMM (gemm_pipe.h) with parts of the tile loop switched off (results wrong, timing
# only): which part bounds it.   usage: tools/pipe_variants.sh noa nodma nolds nosync ... -> tools/_variants/libdc_gp_<v>.so
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/dynamicrafter_amd/csrc"
OUT="$ROOT/tools/_variants"
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CS"
build() {  # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 -c "$CS/gemm_conv_glds.hip" -o "$OUT/gcg_$1.o"
  OBJS=""
  for f in gemm_conv ff_fused norms attention flash_pipe elementwise encoders runtime; do OBJS="$OBJS $CS/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdc_gp_$1.so" $OBJS "$OUT/gcg_$1.o"
  rm -f "$OUT/gcg_$1.o"
}
for v in "$@"; do
  flags=""
  IFS=_ read -ra parts <<< "$v"
  for q in "${parts[@]}"; do
    case "$q" in
      noa) flags="$flags -DGP_DBG_NOA" ;;
      nodma) flags="$flags -DGP_DBG_NODMA" ;;
      nolds) flags="$flags -DGP_DBG_NOLDS" ;;
      nosync) flags="$flags -DGP_DBG_NOSYNC" ;;
      arows) flags="$flags -DGP_DBG_A_ROWS" ;;
      asame) flags="$flags -DGP_DBG_A_SAMELINE" ;;
      stamps) flags="$flags -DGP_STAMPS" ;;
      *) echo "unknown part $q"; exit 1 ;;
    esac
  done
  build "$v" "$flags" &
done
wait
echo built "$@"
