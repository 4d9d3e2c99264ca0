"""Shader clocks per 64-key tile and the clock the chip holds inside flash_attn_d64_pipe_kernel: an instrumented tool build
(-DFP_STAMPS, tools/flash_variants.sh stamps) stamps s_memtime / s_memrealtime (100 MHz) around the tile loop of every
workgroup.   usage (GPU box): python tools/flash_stamps.py [batch heads L]   (after tools/flash_variants.sh stamps)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DC_HIP_LIB", os.path.join(ROOT, "tools", "_variants", "libdc_stamps.so"))
sys.path.insert(0, ROOT)
import torch
from dynamicrafter_amd import ops, _hip
DEV = "cuda:0"
batch, heads, L = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 5, 9216)
C = heads * 64
qkv = torch.randn(batch * L, 3 * C, device=DEV).to(torch.bfloat16)
o = torch.empty(batch * L, C, dtype=torch.bfloat16, device=DEV)
f = lambda: ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=batch, heads=heads, Lq=L, Lk=L, scale=0.125)
dbg = ctypes.CDLL(os.environ["DC_HIP_LIB"]).dc_fp_debug_stamps
dbg.restype = ctypes.c_int; dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 4)()
for _ in range(20): f()
torch.cuda.synchronize(); dbg(buf, 1)
for _ in range(5): f()
torch.cuda.synchronize(); dbg(buf, 0)
clk, rt, tiles, wgs = (int(b) for b in buf)
print(f"flash_pipe {batch}x{heads}x{L}: {clk / tiles:.0f} shader clocks per 64-key tile (32 + 8 MFMAs = 1280 pipe cycles), "
      f"clock {clk / rt * 100:.0f} MHz, {wgs} workgroup passes")
