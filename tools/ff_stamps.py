import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DC_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdc_dbg.so")
from dynamicrafter_amd import ops, _hip
DEV = "cuda:0"; M = 294912
g = torch.Generator().manual_seed(1)
w1 = torch.randn(2560, 320, generator=g) * 320 ** -0.5; b1 = torch.randn(2560, generator=g) * 0.1
w2 = torch.randn(320, 1280, generator=g) * 1280 ** -0.5; b2 = torch.randn(320, generator=g) * 0.1
pw1 = ops.PackedWeight.linear(w1, b1, DEV); pw2 = ops.PackedWeight.linear(w2, b2, DEV); w2p = ops.ff2_permuted(w2, DEV)
x = torch.randn(M, 320, device=DEV).to(torch.bfloat16); h = torch.randn(M, 320, device=DEV).to(torch.bfloat16)
for _ in range(3): ops.ff_geglu_fused320(x, pw1, w2p, pw2.bias, h, residual=h)
torch.cuda.synchronize()
l = C.CDLL(os.environ["DC_HIP_LIB"])
buf = (C.c_ulonglong * 20)()
l.dc_ff_dbg(buf)
names = ["wait+barrier", "dma issue", "phase1(j+1)", "geglu(j)", "phase2(j)"]
for w in range(4):
    v = [buf[w * 5 + i] for i in range(5)]
    print(f"wave {w}: " + "  ".join(f"{n} {x / 40:.0f}" for n, x in zip(names, v)) + f"   sum/chunk {sum(v) / 40:.0f} cycles")
