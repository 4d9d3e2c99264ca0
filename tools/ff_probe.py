"""Fused FeedForward (dim 320) vs the ff1(GEGLU) + ff2 GEMM pair at the level-0 row count. usage: ff_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 294912
g = torch.Generator().manual_seed(1)
w1 = torch.randn(2560, 320, generator=g) * 320 ** -0.5; b1 = torch.randn(2560, generator=g) * 0.1
w2 = torch.randn(320, 1280, generator=g) * 1280 ** -0.5; b2 = torch.randn(320, generator=g) * 0.1
pw1 = ops.PackedWeight.linear(w1, b1, DEV); pw2 = ops.PackedWeight.linear(w2, b2, DEV); w2p = ops.ff2_permuted(w2, DEV)
x = torch.randn(M, 320, device=DEV).to(torch.bfloat16); h = torch.randn(M, 320, device=DEV).to(torch.bfloat16)
mid = torch.empty(M, 1280, dtype=torch.bfloat16, device=DEV)
def pair():
    ops.gemm(x, pw1, mid, geglu=True); ops.gemm(mid, pw2, h, residual=h)
def fused():
    ops.ff_geglu_fused320(x, pw1, w2p, pw2.bias, h, residual=h)
gam = torch.ones(320, device=DEV); bet = torch.zeros(320, device=DEV)
def ln_pair():
    ops.layernorm(h, x, gam, bet, 1e-5); pair()
def ln_fused():
    ops.ff_geglu_fused320(h, pw1, w2p, pw2.bias, h, residual=h, ln=(gam, bet))
for name, fn in (("pair", pair), ("fused", fused), ("ln+pair", ln_pair), ("lnfused", ln_fused), ("pair", pair), ("fused", fused),
                 ("ln+pair", ln_pair), ("lnfused", ln_fused)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{name:8s} M={M}: {us:8.1f} us  {2.0 * M * 3840 * 320 / us / 1e6:7.1f} TF/s", flush=True)

# LayerNorm + Linear (dc_ln_linear) vs LayerNorm kernel + GEMM at the level-0 and level-1 shapes
for (Mr, K, N) in ((M, 320, 960), (M, 320, 320), (M // 4, 640, 1920), (M // 4, 640, 640)):
    pw = ops.PackedWeight.linear(torch.randn(N, K, generator=g) * K ** -0.5, None, DEV)
    hh = torch.randn(Mr, K, device=DEV).to(torch.bfloat16); nn_ = torch.empty_like(hh)
    gk = torch.ones(K, device=DEV); bk = torch.zeros(K, device=DEV)
    o = torch.empty(Mr, N, dtype=torch.bfloat16, device=DEV)
    def two(): ops.layernorm(hh, nn_, gk, bk, 1e-5); ops.gemm(nn_, pw, o)
    def one(): ops.ln_linear(hh, pw, o, ln=(gk, bk))
    for name, fn in (("ln+gemm", two), ("ln_linear", one), ("ln+gemm", two), ("ln_linear", one)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{name:10s} M={Mr} K={K} N={N}: {us:8.1f} us  {2.0 * Mr * N * K / us / 1e6:7.1f} TF/s  {2.0 * Mr * (K + N) / us / 1e6:6.2f} TB/s", flush=True)

# LayerNorm + qkv + temporal attention in one kernel vs the three kernels (level 0: B = 2 clips, HW = 9216)
Bc, T, HW = 2, 16, M // 32
pwq = ops.PackedWeight.linear(torch.randn(960, 320, generator=g) * 320 ** -0.5, None, DEV)
qkv = torch.empty(M, 960, dtype=torch.bfloat16, device=DEV); att = torch.empty(M, 320, dtype=torch.bfloat16, device=DEV)
def three(): ops.ln_linear(h, pwq, qkv, ln=(gam, bet)); ops.temporal_attn(qkv, att, B=Bc, T=T, HW=HW, heads=5, scale=0.125)
def fusedta(): ops.ln_qkv_temporal_attn320(h, (gam, bet), pwq, att, B=Bc, T=T, HW=HW, scale=0.125)
for name, fn in (("ln_qkv+tattn", three), ("fused", fusedta), ("ln_qkv+tattn", three), ("fused", fusedta)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:13s} M={M}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)

# GroupNorm + SiLU + temporal conv in one kernel vs GroupNorm kernels + implicit-GEMM tconv (level 0)
wt = torch.randn(320, 320, 3, 1, 1, generator=g) * 960 ** -0.5
pwt = ops.PackedWeight.tconv3(wt, torch.zeros(320), DEV)
st = torch.empty(Bc * 64, dtype=torch.float32, device=DEV); nn2 = torch.empty_like(h); o2 = torch.empty_like(h)
def gn_tc():
    ops.groupnorm(h, nn2, gam, bet, groups=32, n_inst=Bc, rows_per_inst=T * HW, eps=1e-5, silu=True)
    ops.gemm(nn2, pwt, o2, tconv=dict(T=T, HW=HW))
def fused_tc():
    ops.groupnorm_stats(h, st, groups=32, n_inst=Bc, rows_per_inst=T * HW, eps=1e-5)
    ops.gn_silu_tconv3(h, gam, bet, st, pwt, o2, B=Bc, T=T, HW=HW)
for name, fn in (("gn+tconv", gn_tc), ("fused", fused_tc), ("gn+tconv", gn_tc), ("fused", fused_tc)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:13s} M={M}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)

# the same at level 1 (640 channels, a quarter of the rows)
M1, HW1 = M // 4, HW // 4
h1 = torch.randn(M1, 640, device=DEV).to(torch.bfloat16); g1 = torch.ones(640, device=DEV); b1_ = torch.zeros(640, device=DEV)
wt1 = torch.randn(640, 640, 3, 1, 1, generator=g) * 1920 ** -0.5
pwt1 = ops.PackedWeight.tconv3(wt1, torch.zeros(640), DEV)
nn3 = torch.empty_like(h1); o3 = torch.empty_like(h1)
def gn_tc1():
    ops.groupnorm(h1, nn3, g1, b1_, groups=32, n_inst=Bc, rows_per_inst=T * HW1, eps=1e-5, silu=True)
    ops.gemm(nn3, pwt1, o3, tconv=dict(T=T, HW=HW1))
def fused_tc1():
    ops.groupnorm_stats(h1, st, groups=32, n_inst=Bc, rows_per_inst=T * HW1, eps=1e-5)
    ops.gn_silu_tconv3(h1, g1, b1_, st, pwt1, o3, B=Bc, T=T, HW=HW1)
for name, fn in (("gn+tconv 640", gn_tc1), ("fused 640", fused_tc1), ("gn+tconv 640", gn_tc1), ("fused 640", fused_tc1)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:13s} M={M1}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us", flush=True)
