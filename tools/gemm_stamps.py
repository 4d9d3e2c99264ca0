"""Where does a wave of gemm_conv_glds_kernel spend its K loop? Builds an instrumented copy of the library
(-DDC_GEMM_STAMPS: shader-clock stamps after the vmcnt wait, after the barrier and after the K-step body), runs ONE
shape and prints the per-K-tile averages.   usage (GPU box): python tools/gemm_stamps.py conv|lin|tconv ci co H W
The instrumented kernel is a tool build (DC_HIP_LIB points _hip.py at it); the product library is untouched."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/dc_stamps"
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libdcrafter_hip_stamps.so")
env = dict(os.environ, DC_OUT=lib, DC_OBJDIR=out, DC_EXTRA_FLAGS="-DDC_GEMM_STAMPS " + os.environ.get("STAMPS_EXTRA", ""))
subprocess.check_call([os.path.join(ROOT, "dynamicrafter_amd", "csrc", "build.sh")], env=env, stdout=subprocess.DEVNULL)
os.environ["DC_HIP_LIB"] = lib
os.environ["DC_GEMM_SPLITK"] = "0"
os.environ["DC_GEMM_PERSIST"] = "0"
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
st = ws.view(torch.int64)[: 8 * 4096 * 4].view(-1, 4).cpu()
st = st[st[:, 3] > 0].double()
nk = st[:, 3]
print(f"{kind} M={M} N={co} K={pw.K}: {st.shape[0]} waves, {int(nk[0])} K tiles each")
for i, name in enumerate(("vmcnt wait", "barrier", "K-step body")):
    per = (st[:, i] / nk)
    print(f"  {name:12s} mean {per.mean():8.1f}  min {per.min():8.1f}  max {per.max():8.1f}  shader clocks per K tile")
tot = (st[:, :3].sum(1) / nk).mean()
print(f"  total {tot:.1f} clocks per K tile per wave (MFMA alone: {2 * (co if co <= 320 else 320) // 64 * 4 * 32} per wave, x2 waves per SIMD)")
