"""Run ONE dc_flash_attn_d64 shape a few times (for rocprofv3 --pmc passes). usage: one_flash.py [batch heads Lq Lk]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
batch, heads, Lq, Lk = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 5, 9216, 9216)
C = heads * 64
qkv = torch.randn(batch * Lq, 3 * C, device=DEV).to(torch.bfloat16)
o = torch.empty(batch * Lq, C, dtype=torch.bfloat16, device=DEV)
for _ in range(3):
    ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, batch=batch, heads=heads, Lq=Lq, Lk=Lk, scale=0.125)
torch.cuda.synchronize()
print("done", batch, heads, Lq, Lk)
