"""Run ONE dc_gemm_conv shape a few times (for rocprofv3 --pmc passes). usage: one_gemm.py conv|lin|tconv|geglu"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
from dynamicrafter_amd.ops import PackedWeight
DEV = "cuda:0"
kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
ci = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
co = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
H, W = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (18, 32)
M = 32 * H * W
x = torch.randn(M, ci, device=DEV).to(torch.bfloat16)
if kind == "conv":
    pw = PackedWeight.conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DEV)
    kw = dict(conv=dict(IH=H, IW=W, OH=H, OW=W, stride=1, pad=1, ups=0)); no = co
elif kind == "tconv":
    pw = PackedWeight.tconv3(torch.randn(co, ci, 3, 1, 1) * (3 * ci) ** -0.5, torch.randn(co), DEV)
    kw = dict(tconv=dict(T=16, HW=H * W)); no = co
else:
    pw = PackedWeight.linear(torch.randn(co, ci) * ci ** -0.5, torch.randn(co), DEV)
    kw = dict(geglu=(kind == "geglu")); no = co // 2 if kind == "geglu" else co
out = torch.empty(M, no, dtype=torch.bfloat16, device=DEV)
for _ in range(5):
    ops.gemm(x, pw, out, **kw)
torch.cuda.synchronize()
print("done", kind, M, ci, co)
