#!/bin/bash
# What bounds gemm_pp_kernel? Tool builds with operands aliased (every workgroup reads row panel 0 and / or weight tile 0: all
# operand bytes come from L1 / L2) or the activation loads switched off; results are wrong, only the time matters.
# Build (container): for each variant  tools/build_variant.sh pp_<v> "<flags>" gemm_conv_glds   (flags: -DPP_DBG_ALIAS_A,
# -DPP_DBG_ALIAS_W, -DPP_DBG_NO_A).  usage (GPU box): bash tools/pp_variants.sh
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SH="geglu 640,geglu 1280->10240 L2"
echo "== product build"; DC_GEMM_PLAN=51 python $ROOT/tools/gemm_bench.py --iters 20 --only "$SH" 2>&1 | grep -v amdgpu.ids
for v in alias_aw dummy_a dummy_w dummy_aw no_store dummy_aw_no_store; do
  echo "== $v"; DC_GEMM_PLAN=51 DC_HIP_LIB=$ROOT/tools/_variants/libdc_pp_$v.so python $ROOT/tools/gemm_bench.py --iters 20 --only "$SH" 2>&1 | grep -v amdgpu.ids
done
