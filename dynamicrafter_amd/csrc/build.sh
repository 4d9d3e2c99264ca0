#!/bin/bash
# Builds libdcrafter_hip.so for gfx950 in-tree (next to the sources). hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="${DC_OUT:-$HERE/libdcrafter_hip.so}"      # DC_OUT / DC_EXTRA_FLAGS: instrumented tool builds (tools/gemm_stamps.py)
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$HERE ${DC_EXTRA_FLAGS:-}"
OBJDIR="${DC_OBJDIR:-$HERE}"
OBJS=()
pids=()
SRCS="gemm_conv gemm_conv_glds ff_fused norms attention flash_pipe elementwise encoders runtime"
# objects of kernels that no longer exist must not ride along to the GPU box (or into a link by hand)
for o in "$OBJDIR"/*.o; do
  [ -e "$o" ] || continue
  b="$(basename "$o" .o)"
  case " $SRCS " in *" $b "*) ;; *) echo "removing stale object $o"; rm -f "$o" ;; esac
done
# ff_fused.hip: SLP vectorisation off. hipcc packs adjacent scalar f32 adds / muls / fmas of the X-stationary kernels into v_pk_*_f32,
# and beside MFMAs a packed f32 instruction costs more issue time than the two it replaces (MI355X_MICROARCH 'price of one filler'):
# same-box A/B -0.85 ms per 1024 step; the other files measured no difference (profiles/r04_ff_gelu_slp_ab.txt).
file_flags() { case "$1" in ff_fused) echo "-fno-slp-vectorize" ;; *) echo "" ;; esac; }
for f in $SRCS; do
  "$HIPCC" $FLAGS $(file_flags "$f") -c "$HERE/$f.hip" -o "$OBJDIR/$f.o" &
  pids+=($!)
  OBJS+=("$OBJDIR/$f.o")
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
