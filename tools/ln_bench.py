"""Time dc_layernorm on the level-0 / level-1 token shapes (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
for rows, C in ((294912, 320), (73728, 640), (18432, 1280)):
    x = torch.randn(rows, C, device=DEV).to(torch.bfloat16); y = torch.empty_like(x)
    g = torch.randn(C, device=DEV); b = torch.randn(C, device=DEV)
    for _ in range(3): ops.layernorm(x, y, g, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.layernorm(x, y, g, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"layernorm [{rows} x {C}]: {ms * 1e3:.1f} us  {4.0 * rows * C / ms / 1e6:.0f} GB/s  ")
