#!/bin/bash
# Tool build of libdcrafter_hip.so: recompiles the named kernel files with extra -D flags and links them with the PRODUCT objects
# of the other files (dynamicrafter_amd/csrc/*.o, built by build.sh) into tools/_variants/libdc_<name>.so. Load it with
# DC_HIP_LIB. usage: tools/build_variant.sh <name> "<flags>" <file> [<file> ...]      (files without the .hip suffix)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/dynamicrafter_amd/csrc"
NAME="$1"; FL="$2"; shift 2
OUT="$ROOT/tools/_variants"
mkdir -p "$OUT/obj_$NAME"
SRCS="$(grep '^SRCS=' "$CS/build.sh" | cut -d'"' -f2)"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CS $FL"
OBJS=()
for f in $SRCS; do
  case " $* " in
    *" $f "*) PF=""; [ "$f" = ff_fused ] && [ -z "${DC_VARIANT_SLP:-}" ] && PF="-fno-slp-vectorize"      # the product's per-file flag (build.sh)
              /opt/rocm/bin/hipcc $FLAGS $PF -c "$CS/$f.hip" -o "$OUT/obj_$NAME/$f.o" & OBJS+=("$OUT/obj_$NAME/$f.o") ;;
    *) [ -e "$CS/$f.o" ] || { echo "missing $CS/$f.o: run dynamicrafter_amd/csrc/build.sh first" >&2; exit 1; }
       OBJS+=("$CS/$f.o") ;;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdc_$NAME.so" "${OBJS[@]}"
echo "built $OUT/libdc_$NAME.so"
