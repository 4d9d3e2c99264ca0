#!/bin/bash
# Tool builds of flash_pipe.hip with parts of the main loop switched off (results wrong, timing only): which part bounds it.
# usage: tools/flash_variants.sh   -> tools/_variants/libdc_<variant>.so, select with DC_HIP_LIB
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/dynamicrafter_amd/csrc"
OUT="$ROOT/tools/_variants"
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CS"
build() {  # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 -c "$CS/flash_pipe.hip" -o "$OUT/flash_pipe_$1.o"
  OBJS=""
  for f in gemm_conv gemm_conv_glds ff_fused norms attention elementwise encoders runtime; do OBJS="$OBJS $CS/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdc_$1.so" $OBJS "$OUT/flash_pipe_$1.o"
  rm -f "$OUT/flash_pipe_$1.o"
}
for v in "$@"; do
  case "$v" in
    noex) build noex "-DFP_DBG_NOEX" ;;
    nolds) build nolds "-DFP_DBG_NOLDS" ;;
    nostage) build nostage "-DFP_DBG_NOSTAGE" ;;
    nobar) build nobar "-DFP_DBG_NOSTAGE -DFP_DBG_NOBAR" ;;
    stamps) build stamps "-DFP_STAMPS" ;;
    st_noex) build st_noex "-DFP_STAMPS -DFP_DBG_NOEX" ;;
    st_nolds) build st_nolds "-DFP_STAMPS -DFP_DBG_NOLDS" ;;
    st_nostage) build st_nostage "-DFP_STAMPS -DFP_DBG_NOSTAGE -DFP_DBG_NOBAR" ;;
    st_nolds_nostage) build st_nolds_nostage "-DFP_STAMPS -DFP_DBG_NOLDS -DFP_DBG_NOSTAGE -DFP_DBG_NOBAR" ;;
    st_mfma) build st_mfma "-DFP_STAMPS -DFP_DBG_NOEX -DFP_DBG_NOLDS -DFP_DBG_NOSTAGE -DFP_DBG_NOBAR" ;;
    st_nobar) build st_nobar "-DFP_STAMPS -DFP_DBG_NOBAR" ;;
    st_hwbar) build st_hwbar "-DFP_STAMPS -DFP_HW_BARRIER" ;;
    st_stag2) build st_stag2 "-DFP_STAMPS -DFP_STAGGER=2" ;;
    st_stag4) build st_stag4 "-DFP_STAMPS -DFP_STAGGER=4" ;;
    st_stag8) build st_stag8 "-DFP_STAMPS -DFP_STAGGER=8" ;;
    st_noload) build st_noload "-DFP_STAMPS -DFP_DBG_NOLOAD" ;;
    st_nostore) build st_nostore "-DFP_STAMPS -DFP_DBG_NOSTORE" ;;
    st_vinstep) build st_vinstep "-DFP_STAMPS -DFP_V_IN_STEP" ;;
    st_noex_nostage) build st_noex_nostage "-DFP_STAMPS -DFP_DBG_NOEX -DFP_DBG_NOSTAGE -DFP_DBG_NOBAR" ;;
    x16_*) # tool builds of flash_pipe16.hip: x16_nolds, x16_nostage, x16_noex, x16_nolds_nostage, x16_mfma
      fl=""; case "$v" in *nolds*) fl="$fl -DF16_DBG_NOLDS";; esac; case "$v" in *nostage*) fl="$fl -DF16_DBG_NOSTAGE";; esac
      case "$v" in *noex*) fl="$fl -DF16_DBG_NOEX";; esac; case "$v" in *mfma*) fl="-DF16_DBG_NOLDS -DF16_DBG_NOSTAGE -DF16_DBG_NOEX";; esac
      /opt/rocm/bin/hipcc $FLAGS $fl -c "$CS/flash_pipe16.hip" -o "$OUT/fp16_$v.o"
      OBJS=""
      for f in gemm_conv gemm_conv_glds ff_fused norms attention flash_pipe elementwise encoders runtime; do OBJS="$OBJS $CS/$f.o"; done
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libdc_$v.so" $OBJS "$OUT/fp16_$v.o"
      rm -f "$OUT/fp16_$v.o" ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
  echo "built $v"
done
