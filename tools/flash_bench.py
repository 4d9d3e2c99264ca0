"""Time dc_flash_attn_d64 on the level-0 self-attention shape (HIP events). usage: flash_bench.py [batch heads L]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
batch, heads, L = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 5, 9216)
C = heads * 64
qkv = torch.randn(batch * L, 3 * C, device=DEV).to(torch.bfloat16)
o = torch.empty(batch * L, C, dtype=torch.bfloat16, device=DEV)
f = lambda: ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=batch, heads=heads, Lq=L, Lk=L, scale=0.125)
for _ in range(2): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"flash {batch}x{heads}x{L}: {ms:.3f} ms  {4 * batch * heads * L * L * 64 / ms / 1e9:.0f} TF/s  (DC_FLASH_PP={os.environ.get('DC_FLASH_PP', '1')})")
