"""Where does a wave of gemm_pipe320_kernel / conv3_pipe320_kernel wait, and what clock does the chip hold under it? Builds an instrumented copy of the library (-DGP_STAMPS: shader clocks at the
four waiting points of the tile loop), runs ONE conv / linear shape as whole tiles and prints per-K-tile averages.
usage (GPU box): python tools/pipe_stamps.py conv|lin|tconv ci co H W      (STAMPS_EXTRA: more -D flags, e.g. -DGP_DBG_NOA)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/dc_pstamps"
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libdcrafter_hip_stamps.so")
env = dict(os.environ, DC_OUT=lib, DC_OBJDIR=out, DC_EXTRA_FLAGS="-DGP_STAMPS " + os.environ.get("STAMPS_EXTRA", ""))
subprocess.check_call([os.path.join(ROOT, "dynamicrafter_amd", "csrc", "build.sh")], env=env, stdout=subprocess.DEVNULL)
os.environ["DC_HIP_LIB"] = lib
os.environ["DC_GEMM_SPLITK"] = "0"
os.environ["DC_GEMM_PERSIST"] = "0"
os.environ["DC_GEMM_TILE"] = "320"
os.environ.setdefault("DC_GEMM_PLAN", "1")       # 1: gemm_pipe.h, 5: conv_pipe.h for the stride-1 3x3 convs
sys.path.insert(0, ROOT)
import torch
from dynamicrafter_amd import ops
from dynamicrafter_amd.ops import PackedWeight
DEV = "cuda:0"
kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
ci, co, H, W = (int(a) for a in sys.argv[2:6]) if len(sys.argv) > 5 else (640, 640, 36, 64)
M = 32 * H * W
x = torch.randn(M, ci, device=DEV).to(torch.bfloat16)
if kind == "conv":
    pw = PackedWeight.conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DEV)
    kw = dict(conv=dict(IH=H, IW=W, OH=H, OW=W, stride=1, pad=1, ups=0))
elif kind == "tconv":
    pw = PackedWeight.tconv3(torch.randn(co, ci, 3, 1, 1) * (3 * ci) ** -0.5, torch.randn(co), DEV)
    kw = dict(tconv=dict(T=16, HW=H * W))
else:
    pw = PackedWeight.linear(torch.randn(co, ci) * ci ** -0.5, torch.randn(co), DEV)
    kw = {}
o = torch.empty(M, co, dtype=torch.bfloat16, device=DEV)
ws = ops._gemm_workspace(torch.device(DEV), torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    ws.zero_()
    ops.gemm(x, pw, o, **kw)
torch.cuda.synchronize()
print(ops._hip.lib().dc_gemm_last_variant().decode())
st = ws.view(torch.int64)[: 8 * 8192].view(-1, 8).cpu()
st = st[st[:, 5] > 0].double()
nk = st[:, 5]
print(f"{kind} M={M} N={co} K={pw.K}: {st.shape[0]} waves, {int(nk[0])} K tiles each")
for i, name in enumerate(("activations (vmcnt)", "own weight pieces (vmcnt)", "stage free (counter)", "weights landed (counter)", "whole loop")):
    per = st[:, i] / nk
    print(f"  {name:26s} mean {per.mean():8.1f}  min {per.min():8.1f}  max {per.max():8.1f}  shader clocks per K tile")
print("  MFMA alone: 2560 clocks per K tile (80 x 32)")
ghz = (st[:, 4] / st[:, 6]) * 0.1
print(f"  shader clock during the loop (s_memtime / s_memrealtime): mean {ghz.mean():.2f} GHz  min {ghz.min():.2f}  max {ghz.max():.2f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gemm(x, pw, o, **kw)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
fl = 2.0 * M * co * pw.K
print(f"  {us:.1f} us per launch (stamped build), {fl / us / 1e6:.1f} TF/s")
