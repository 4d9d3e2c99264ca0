"""Time dc_cross_attn_dual_d64 on the level-0 / 1 / 2 shapes of the 1024 config (32 frames, 77 text + 16 image tokens).
usage: python tools/one_xattn.py        (DC_XATTN_RESIDENT=0: the tile-by-tile kernel)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
for heads, Lq in ((5, 9216), (10, 2304), (20, 576)):
    B, nt, ni = 32, 77, 16
    C, Lc = heads * 64, nt + ni
    q = torch.randn(B * Lq, C, device=DEV).to(torch.bfloat16)
    kv = torch.randn(B * Lc, 4 * C, device=DEV).to(torch.bfloat16)
    o = torch.empty(B * Lq, C, dtype=torch.bfloat16, device=DEV)
    def run():
        ops.cross_attn_dual(q, kv[:, :C], kv[:, C:2 * C], kv[nt:, 2 * C:3 * C], kv[nt:, 3 * C:], o, batch=B, heads=heads, Lq=Lq, Lk=nt,
                            Lk2=ni, scale=0.125, scale2=1.0, kv_bstride=Lc)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    gb = 2 * B * Lq * C * 2 / 1e9
    print(f"cross-attention dual  heads {heads:2d}  Lq {Lq:5d}: {us:7.1f} us  ({gb / us * 1e6 / 1e3:.2f} TB/s of q + o)")
