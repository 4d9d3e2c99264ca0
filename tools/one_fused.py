"""Run ONE fused level-0 kernel a few times (for rocprofv3 --pmc passes through tools/pmc_one_gemm.sh with
PMC_SCRIPT=one_fused.py). usage: one_fused.py ff | tconv | lnlin | linres | tattn | tattn640   [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
DEV = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "ff"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 294912
g = torch.Generator().manual_seed(1)
h = torch.randn(M, 320, device=DEV).to(torch.bfloat16); x = torch.randn(M, 320, device=DEV).to(torch.bfloat16)
gam = torch.ones(320, device=DEV); bet = torch.zeros(320, device=DEV)
Bc, T, HW = 2, 16, M // 32
if which == "ff":
    w1 = torch.randn(2560, 320, generator=g) * 320 ** -0.5; b1 = torch.randn(2560, generator=g) * 0.1
    w2 = torch.randn(320, 1280, generator=g) * 1280 ** -0.5; b2 = torch.randn(320, generator=g) * 0.1
    wp = torch.randn(320, 320, generator=g) * 320 ** -0.5
    pw1 = ops.PackedWeight.linear(w1, b1, DEV); pw2 = ops.PackedWeight.linear(w2, b2, DEV); w2p = ops.ff2_permuted(w2, DEV)
    wpp = ops.ff2_permuted(wp, DEV); o = torch.empty_like(h)
    fn = lambda: ops.ff_geglu_proj_fused320(h, pw1, w2p, pw2.bias, wpp, b2.to(DEV), x, o, ln=(gam, bet))
elif which == "tconv":
    wt = torch.randn(320, 320, 3, 1, 1, generator=g) * 960 ** -0.5
    pwt = ops.PackedWeight.tconv3(wt, torch.zeros(320), DEV)
    st = torch.empty(Bc * 64, dtype=torch.float32, device=DEV); o = torch.empty_like(h)
    ops.groupnorm_stats(h, st, groups=32, n_inst=Bc, rows_per_inst=T * HW, eps=1e-5)
    fn = lambda: ops.gn_silu_tconv3(h, gam, bet, st, pwt, o, B=Bc, T=T, HW=HW, residual=x)
elif which == "lnlin":
    pw = ops.PackedWeight.linear(torch.randn(960, 320, generator=g) * 320 ** -0.5, None, DEV)
    o = torch.empty(M, 960, dtype=torch.bfloat16, device=DEV)
    fn = lambda: ops.ln_linear(h, pw, o, ln=(gam, bet))
elif which == "linres":
    pw = ops.PackedWeight.linear(torch.randn(320, 320, generator=g) * 320 ** -0.5, torch.randn(320, generator=g) * 0.1, DEV)
    fn = lambda: ops.linear_residual(x, pw, h, h)
elif which == "tattn":
    pwq = ops.PackedWeight.linear(torch.randn(960, 320, generator=g) * 320 ** -0.5, None, DEV)
    o = torch.empty_like(h)
    fn = lambda: ops.ln_qkv_temporal_attn320(h, (gam, bet), pwq, o, B=Bc, T=T, HW=HW, scale=0.125)
elif which == "tattn640":                     # level 1: dim 640, M / 4 rows
    M4 = M // 4
    h6 = torch.randn(M4, 640, device=DEV).to(torch.bfloat16)
    pwq = ops.PackedWeight.linear(torch.randn(1920, 640, generator=g) * 640 ** -0.5, None, DEV)
    o = torch.empty_like(h6)
    g6 = torch.ones(640, device=DEV); b6 = torch.zeros(640, device=DEV)
    fn = lambda: ops.ln_qkv_temporal_attn(h6, (g6, b6), pwq, o, B=Bc, T=T, HW=M4 // 32, scale=0.125)
else:
    raise SystemExit(f"unknown kernel {which}")
for _ in range(3): fn()
torch.cuda.synchronize()
if os.environ.get("ONE_TIME"):                  # HIP-event time of 20 launches (same-box A/B of tool builds through DC_HIP_LIB)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{which} M={M} {os.environ.get('DC_HIP_LIB', 'product')}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
print("done", which, M)
