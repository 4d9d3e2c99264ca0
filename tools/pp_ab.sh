#!/bin/bash
# Same-box A/B of the GEGLU projections: persistent 8-wave kernel (plan 3) with the erf and the polynomial GELU, and the
# ping-pong kernel (plan 19). usage (GPU box): bash tools/pp_ab.sh        (needs tools/_variants/libdc_gelu_erf.so)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SH="geglu 640,geglu 1280"
echo "== persistent 8-wave kernel, erf GELU (tool build -DDC_GELU_ERF, DC_GEMM_PLAN=3)"
DC_HIP_LIB=$ROOT/tools/_variants/libdc_gelu_erf.so DC_GEMM_PLAN=3 python $ROOT/tools/gemm_bench.py --iters 20 --only "$SH" 2>&1 | grep -v amdgpu.ids
echo "== persistent 8-wave kernel, polynomial GELU (DC_GEMM_PLAN=3)"
DC_GEMM_PLAN=3 python $ROOT/tools/gemm_bench.py --iters 20 --only "$SH" 2>&1 | grep -v amdgpu.ids
echo "== ping-pong kernel (DC_GEMM_PLAN=51)"
DC_GEMM_PLAN=51 python $ROOT/tools/gemm_bench.py --iters 20 --only "$SH" 2>&1 | grep -v amdgpu.ids
echo "== ping-pong kernel, erf GELU (tool build, DC_GEMM_PLAN=51)"
DC_HIP_LIB=$ROOT/tools/_variants/libdc_gelu_erf.so DC_GEMM_PLAN=51 python $ROOT/tools/gemm_bench.py --iters 20 --only "$SH" 2>&1 | grep -v amdgpu.ids
