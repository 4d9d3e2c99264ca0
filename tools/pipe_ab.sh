#!/bin/bash
# Same-box comparison of the three 320-wide conv plans (dc_gemm_set_plan / DC_GEMM_PLAN) and of the pipe GEMM tool builds
# (tools/pipe_variants.sh) on a few conv shapes.  usage: tools/pipe_ab.sh [v1 v2 ...]   (PIPE_PLAN: plan for the tool builds)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SH="${PIPE_SHAPES:-conv3x3 640->640,conv3x3 1280->1280 18x32,conv3x3 1920->640,conv3x3 960->320}"
echo "== 8-wave kernels (DC_GEMM_PLAN=0)"; DC_GEMM_PLAN=0 python $ROOT/tools/gemm_bench.py --iters 10 --only "$SH" 2>&1 | grep -v amdgpu.ids
echo "== gemm_pipe (DC_GEMM_PLAN=1)"; DC_GEMM_PLAN=1 python $ROOT/tools/gemm_bench.py --iters 10 --only "$SH" 2>&1 | grep -v amdgpu.ids
echo "== gemm_pipe on 16x16x32 (DC_GEMM_PLAN=9)"; DC_GEMM_PLAN=9 python $ROOT/tools/gemm_bench.py --iters 10 --only "$SH" 2>&1 | grep -v amdgpu.ids
echo "== conv_pipe (DC_GEMM_PLAN=5)"; DC_GEMM_PLAN=5 python $ROOT/tools/gemm_bench.py --iters 10 --only "$SH" 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  echo "== $v"; DC_GEMM_PLAN=${PIPE_PLAN:-1} DC_HIP_LIB=$ROOT/tools/_variants/libdc_gp_$v.so python $ROOT/tools/gemm_bench.py --iters 10 --only "$SH" 2>&1 | grep -v amdgpu.ids
done
