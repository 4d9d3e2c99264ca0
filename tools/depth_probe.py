"""Probe: what is the LDS ring depth worth? The same 128-wide persistent GEMM on a 3-deep and a 2-deep ring
(DC_GEMM_STAGES=2), HBM-sourced A vs cache-resident A (all rows aliased). usage: depth_probe.py (run twice with the env)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DC_GEMM_TILE"] = os.environ.get("DC_GEMM_TILE", "0")
from dynamicrafter_amd import ops
from dynamicrafter_amd.ops import PackedWeight
DEV = "cuda:0"

def run(M, N, K, alias, reps=20, res=False):
    pw = PackedWeight.linear(torch.randn(N, K) * K ** -0.5, torch.randn(N), DEV)
    x = torch.randn(1, K, device=DEV).to(torch.bfloat16).expand(M, K) if alias else torch.randn(M, K, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    kw = dict(residual=torch.randn(M, N, device=DEV).to(torch.bfloat16)) if res else {}
    for _ in range(3): ops.gemm(x, pw, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(x, pw, out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms * 1e3, 2.0 * M * N * K / ms / 1e9

for (M, N, K) in [(294912, 384, 320), (294912, 128, 320), (73728, 768, 640), (18432, 1280, 1280), (294912, 512, 512)]:
    for alias in (False, True):
        us, tf = run(M, N, K, alias)
        us2, tf2 = run(M, N, K, alias, res=True)
        print(f"stages={os.environ.get('DC_GEMM_STAGES','3')} M={M} N={N} K={K} alias={alias}: {us:.1f} us {tf:.0f} TF/s | +residual {us2:.1f} us {tf2:.0f} TF/s", flush=True)
