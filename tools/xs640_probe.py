"""The X-stationary Linear kernels at K = 640 (level 1 of the 1024 config): ln_linear [73728 x 1920 / 640 x 640], linear_residual
[73728 x 640 x 640], gn_linear [73728 x 640 x 640].   usage: python tools/xs640_probe.py   (DC_HIP_LIB: another build for A/B)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
g = torch.Generator().manual_seed(2)
M, K = 73728, 640
x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV); r = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
gam, bet = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
st = torch.empty(32 * 64, dtype=torch.float32, device=DEV)
ops.groupnorm_stats(x, st, groups=32, n_inst=32, rows_per_inst=M // 32, eps=1e-6)
cases = []
for N in (1920, 640):
    pw = ops.PackedWeight.linear(torch.randn(N, K, generator=g) * K ** -0.5, None, DEV)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    cases.append((f"ln_linear N={N}", lambda pw=pw, o=o: ops.ln_linear(x, pw, o, ln=(gam, bet)), N))
pwb = ops.PackedWeight.linear(torch.randn(640, K, generator=g) * K ** -0.5, torch.zeros(640), DEV)
o2 = torch.empty(M, 640, dtype=torch.bfloat16, device=DEV)
cases.append(("linear_residual N=640", lambda: ops.linear_residual(x, pwb, r, o2), 640))
cases.append(("gn_linear N=640", lambda: ops.gn_linear(x, gam, bet, st, pwb, o2, groups=32, rows_per_inst=M // 32), 640))
for name, fn, N in cases * 2:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"{name:24s}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
