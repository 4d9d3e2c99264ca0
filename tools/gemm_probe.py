"""Probe: same GEMM with A rows all aliasing one row (lda=0: operands cache-resident) vs real strides.
Separates memory-side limits from in-core schedule limits. usage: gemm_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
from dynamicrafter_amd.ops import PackedWeight
DEV = "cuda:0"

def run(M, N, K, alias, reps=20, res=False):
    pw = PackedWeight.linear(torch.randn(N, K) * K ** -0.5, torch.randn(N), DEV)
    if alias:
        x = torch.randn(1, K, device=DEV).to(torch.bfloat16).expand(M, K)
    else:
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    kw = dict(residual=torch.randn(M, N, device=DEV).to(torch.bfloat16)) if res else {}
    for _ in range(3): ops.gemm(x, pw, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = ops.current_stream() if hasattr(ops, "current_stream") else None
    e0.record()
    for _ in range(reps): ops.gemm(x, pw, out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * N * K / ms / 1e9

shapes = [(73728, 640, 640), (294912, 320, 320), (18432, 1280, 1280), (73728, 1920, 640), (294912, 960, 320),
          (294912, 320, 1280), (73728, 640, 2560), (18432, 3840, 1280), (73728, 1280, 1280)]
aliases = (False, True) if os.environ.get("PROBE_ALIAS") else (False,)
for (M, N, K) in shapes:
    for alias in aliases:
        ms, tf = run(M, N, K, alias)
        ms2, tf2 = run(M, N, K, alias, res=True)
        print(f"M={M} N={N} K={K} alias={alias}: {ms*1e3:.1f} us {tf:.0f} TF/s | +residual {ms2*1e3:.1f} us {tf2:.0f} TF/s", flush=True)
