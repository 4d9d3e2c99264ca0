"""Micro-benchmark of dc_gemm_conv on the UNet's dominant shapes (batch-2 CFG rows).
usage: python tools/gemm_bench.py [--iters 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops  # noqa: E402
from dynamicrafter_amd.ops import PackedWeight  # noqa: E402

DEV = "cuda:0"
F = 32  # frames (B=2 x T=16)
SHAPES = [
    # name, kind, M or (H,W), Cin, Cout
    ("lin 320x320 L0", "lin", 294912, 320, 320),
    ("lin 640x640 L1", "lin", 73728, 640, 640),
    ("lin 1280x1280 L2", "lin", 18432, 1280, 1280),
    ("ffout 1280->320 L0", "lin", 294912, 1280, 320),
    ("geglu 320->2560 L0", "geglu", 294912, 320, 2560),
    ("geglu 640->5120 L1", "geglu", 73728, 640, 5120),
    ("geglu 1280->10240 L2", "geglu", 18432, 1280, 10240),
    ("geglu 1280->10240 L3", "geglu", 4608, 1280, 10240),
    ("conv3x3 320->320 72x128", "conv", (72, 128), 320, 320),
    ("conv3x3 640->640 36x64", "conv", (36, 64), 640, 640),
    ("conv3x3 1280->1280 18x32", "conv", (18, 32), 1280, 1280),
    ("conv3x3 2560->1280 18x32", "conv", (18, 32), 2560, 1280),
    ("conv3x3 1280->1280 9x16", "conv", (9, 16), 1280, 1280),
    ("conv3x3 640->320 72x128", "conv", (72, 128), 640, 320),
    ("conv3x3 960->320 72x128", "conv", (72, 128), 960, 320),
    ("conv3x3 320->640 36x64", "conv", (36, 64), 320, 640),
    ("conv3x3 1280->640 36x64", "conv", (36, 64), 1280, 640),
    ("conv3x3 1920->640 36x64", "conv", (36, 64), 1920, 640),
    ("conv3x3 640->1280 18x32", "conv", (18, 32), 640, 1280),
    ("conv3x3 1920->1280 18x32", "conv", (18, 32), 1920, 1280),
    ("down3x3 s2 320 72x128", "down", (72, 128), 320, 320),
    ("down3x3 s2 640 36x64", "down", (36, 64), 640, 640),
    ("down3x3 s2 1280 18x32", "down", (18, 32), 1280, 1280),
    ("lin 1280x1280 L3", "lin", 4608, 1280, 1280),
    ("tconv 640 36x64", "tconv", (36, 64), 640, 640),
    ("lin 2560->1280 L2", "lin", 18432, 2560, 1280),
    ("lin 5120->1280 L2", "lin", 18432, 5120, 1280),
    ("tconv 320 72x128", "tconv", (72, 128), 320, 320),
    ("tconv 1280 18x32", "tconv", (18, 32), 1280, 1280),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="substring filter on the shape names")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "ones", "small"],
                    help="operand values: the chip is power-limited under MFMA load and the clock it holds depends on how many bits toggle")
    args = ap.parse_args()
    for name, kind, m, ci, co in SHAPES:
        if args.only and not any(o in name for o in args.only.split(",")):
            continue
        if kind in ("lin", "geglu"):
            M = m
            x = torch.randn(M, ci, device=DEV).to(torch.bfloat16)
            pw = PackedWeight.linear(torch.randn(co, ci) * ci ** -0.5, torch.randn(co), DEV)
            out = torch.empty(M, co // 2 if kind == "geglu" else co, dtype=torch.bfloat16, device=DEV)
            kw = dict(geglu=(kind == "geglu"))
            flops = 2.0 * M * co * ci
        elif kind == "down":                         # Downsample: 3x3, stride 2, pad 1 (input H x W -> H/2 x W/2)
            H, W = m
            Mi, M = F * H * W, F * (H // 2) * (W // 2)
            x = torch.randn(Mi, ci, device=DEV).to(torch.bfloat16)
            pw = PackedWeight.conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DEV)
            out = torch.empty(M, co, dtype=torch.bfloat16, device=DEV)
            kw = dict(conv=dict(IH=H, IW=W, OH=H // 2, OW=W // 2, stride=2, pad=1, ups=0))
            flops = 2.0 * M * co * ci * 9
        elif kind == "conv":
            H, W = m
            M = F * H * W
            x = torch.randn(M, ci, device=DEV).to(torch.bfloat16)
            pw = PackedWeight.conv3x3(torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5, torch.randn(co), DEV)
            out = torch.empty(M, co, dtype=torch.bfloat16, device=DEV)
            kw = dict(conv=dict(IH=H, IW=W, OH=H, OW=W, stride=1, pad=1, ups=0))
            flops = 2.0 * M * co * ci * 9
        else:
            H, W = m
            M = F * H * W
            x = torch.randn(M, ci, device=DEV).to(torch.bfloat16)
            pw = PackedWeight.tconv3(torch.randn(co, ci, 3, 1, 1) * (3 * ci) ** -0.5, torch.randn(co), DEV)
            out = torch.empty(M, co, dtype=torch.bfloat16, device=DEV)
            kw = dict(tconv=dict(T=16, HW=H * W))
            flops = 2.0 * M * co * ci * 3
        if args.data != "randn":
            # same kernels, same addresses, other VALUES (results are not checked here)
            fill = {"zeros": 0.0, "ones": 1.0, "small": 1.0}[args.data]
            x.fill_(fill)
            if args.data == "small":
                x.copy_((torch.randint(0, 2, x.shape, device=DEV) * 2 - 1).to(torch.bfloat16))     # +-1: sign bit only
            if args.data == "zeros":
                pw.w.zero_()
        for _ in range(3):
            ops.gemm(x, pw, out, **kw)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):                       # best of three event-timed batches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                ops.gemm(x, pw, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / args.iters)
        dt = best
        print(f"{name:28s} M={M:7d} {dt * 1e6:9.1f} us  {flops / dt / 1e12:7.1f} TF/s  {ops._hip.lib().dc_gemm_last_variant().decode()}", flush=True)


if __name__ == "__main__":
    main()
