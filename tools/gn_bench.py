"""GroupNorm timing on the UNet's level-0/1 shapes (4-D per-frame and 5-D per-clip statistics). usage: gn_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
for (n_inst, rpi, C) in [(32, 9216, 320), (2, 147456, 320), (32, 2304, 640), (2, 36864, 640), (32, 576, 1280), (2, 9216, 1280)]:
    x = torch.randn(n_inst * rpi, C, device=DEV).to(torch.bfloat16)
    y = torch.empty_like(x)
    g = torch.ones(C, device=DEV); b = torch.zeros(C, device=DEV)
    for _ in range(3): ops.groupnorm(x, y, g, b, groups=32, n_inst=n_inst, rows_per_inst=rpi, eps=1e-5, silu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.groupnorm(x, y, g, b, groups=32, n_inst=n_inst, rows_per_inst=rpi, eps=1e-5, silu=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"gn n_inst={n_inst} rows={rpi} C={C}: {us:.1f} us  {6.0 * n_inst * rpi * C / us / 1e6:.2f} TB/s (3 passes)", flush=True)
    st = torch.empty(n_inst * 64, dtype=torch.float32, device=DEV)
    for _ in range(3): ops.groupnorm_stats(x, st, groups=32, n_inst=n_inst, rows_per_inst=rpi, eps=1e-5)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): ops.groupnorm_stats(x, st, groups=32, n_inst=n_inst, rows_per_inst=rpi, eps=1e-5)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"   statistics only (partial + finalize): {us:.1f} us  {2.0 * n_inst * rpi * C / us / 1e6:.2f} TB/s (1 pass)", flush=True)
