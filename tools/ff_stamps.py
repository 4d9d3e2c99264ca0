"""Where does a wave of ff_geglu_fused320_kernel spend a chunk iteration? Builds an instrumented copy of the library
(-DDC_FF_STAMPS: shader-clock stamps after the DMA wait + barrier, after phase 2 (with the LDS-DMA pieces behind its MFMAs)
and after phase 1 + GEGLU; every stamp drains the LDS queue first, so the parts do not overlap as they do in the product build), runs the
level-0 row count and prints the averages per chunk for wave 0 of workgroup 0.   usage (GPU box): python tools/ff_stamps.py
The instrumented kernel is a tool build (DC_HIP_LIB points _hip.py at it); the product library is untouched."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/dc_ff_stamps"
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libdcrafter_hip_stamps.so")
env = dict(os.environ, DC_OUT=lib, DC_OBJDIR=out, DC_EXTRA_FLAGS="-DDC_FF_STAMPS")
subprocess.check_call([os.path.join(ROOT, "dynamicrafter_amd", "csrc", "build.sh")], env=env, stdout=subprocess.DEVNULL)
os.environ["DC_HIP_LIB"] = lib
sys.path.insert(0, ROOT)
import torch
from dynamicrafter_amd import ops, _hip
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 294912
g = torch.Generator().manual_seed(1)
w1 = torch.randn(2560, 320, generator=g) * 320 ** -0.5; b1 = torch.randn(2560, generator=g) * 0.1
w2 = torch.randn(320, 1280, generator=g) * 1280 ** -0.5; b2 = torch.randn(320, generator=g) * 0.1
pw1 = ops.PackedWeight.linear(w1, b1, DEV); pw2 = ops.PackedWeight.linear(w2, b2, DEV); w2p = ops.ff2_permuted(w2, DEV)
x = torch.randn(M, 320, device=DEV).to(torch.bfloat16); h = torch.randn(M, 320, device=DEV).to(torch.bfloat16)
dbg = _hip.lib().dc_ff_debug_stamps
dbg.restype = ctypes.c_int; dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
ops.ff_geglu_fused320(x, pw1, w2p, pw2.bias, h, residual=h); torch.cuda.synchronize()
dbg(buf, 1)
ops.ff_geglu_fused320(x, pw1, w2p, pw2.bias, h, residual=h); torch.cuda.synchronize()
dbg(buf, 0)
tiles = (M + 127) // 128
n = ((tiles + 255) // 256 if tiles > 256 else 1) * 39          # chunk iterations with all four parts, workgroup 0
names = ("DMA wait + barrier", "(unused)", "phase 1 + GEGLU", "phase 2 + LDS-DMA issue", "(loop overhead)")
tot = 0
for i, nm in enumerate(names):
    if i in (1, 4): continue
    print(f"  {nm:20s} {buf[i] / n:8.1f} shader clocks per chunk"); tot += buf[i] / n
print(f"  total {tot:.1f} per chunk (60 MFMAs = 1920 at 32 cycles each)")
