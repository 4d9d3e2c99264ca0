"""LayerNorm + q/k/v + temporal attention in one kernel vs the three kernels, at the level-0 (dim 320) and level-1 (dim 640) shapes of
the 1024 config (2 clips x 16 frames).   usage (GPU box): python tools/ta_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
for C, HW in ((320, 9216), (640, 2304)):
    B, T, heads = 2, 16, C // 64
    M = B * T * HW
    h = torch.randn(M, C, generator=g).to(torch.bfloat16).to(DEV)
    gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    pwq = ops.PackedWeight.linear(torch.randn(3 * C, C, generator=g) * C ** -0.5, None, DEV)
    qkv = torch.empty(M, 3 * C, dtype=torch.bfloat16, device=DEV); att = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
    def three(): ops.ln_linear(h, pwq, qkv, ln=(gam, bet)); ops.temporal_attn(qkv, att, B=B, T=T, HW=HW, heads=heads, scale=0.125)
    def fused(): ops.ln_qkv_temporal_attn(h, (gam, bet), pwq, att, B=B, T=T, HW=HW, scale=0.125)
    for name, fn in (("ln_qkv+tattn", three), ("fused", fused), ("ln_qkv+tattn", three), ("fused", fused)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"dim {C} M={M} {name:13s}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
