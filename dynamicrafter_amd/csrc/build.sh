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
for f in $SRCS; do
  "$HIPCC" $FLAGS -c "$HERE/$f.hip" -o "$OBJDIR/$f.o" &
  pids+=($!)
  OBJS+=("$OBJDIR/$f.o")
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${OBJS[@]}"
echo "built $OUT"
