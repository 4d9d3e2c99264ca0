#!/usr/bin/env python3
"""bench.py — DynamiCrafter denoising loop on MI355X: denoising-step latency and frames/s, 16 frames @ 576x1024.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2] / SURVEY §8d config 3): inference_1024_v1.0.yaml — latent 16x72x128, DDIM 50
steps `uniform_trailing`, eta 1, CFG 7.5 (cond+uncond evaluated as one batch-2 UNet forward), guidance-rescale 0.7,
v-parameterisation + zero-terminal-SNR + dynamic rescale, fs 10; bf16 weights/activations, fp32 accumulate;
random-init weights and synthetic conditioning (no checkpoints/datasets offline). One clip per GPU (weak scaling,
no data-path collective: conditioning is broadcast once from rank 0 over RCCL before the loop).

A "step" = one captured hipGraph launch = 2 UNet forwards (batched) + fused DDIM update for one clip.
value = frames/s of the whole job = N * 16 / (50 * step + AE encode + AE decode), all three measured here.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STEP_TFLOP = {"1024": 104.672, "512": 25.206, "256": 9.807}        # 2 UNet forwards, 2*MAC
FWD_TFLOP = {"1024": 52.3362, "512": 12.6028, "256": 4.9035}
LATENT = {"1024": (72, 128), "512": (40, 64), "256": (32, 32)}
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


_T0 = time.perf_counter()


def log(msg):
    """progress to stderr (gpurun kills a silent command after 7 minutes)"""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def build_model(res, device):
    import yaml
    from dynamicrafter_amd.utils.utils import instantiate_from_config
    cfg = yaml.safe_load(open(os.path.join(ROOT, "dynamicrafter_amd", "configs", f"inference_{res}_v1.0.yaml")))
    p = cfg["model"]["params"]
    # CLIP towers / Resampler produce the conditioning once per clip and are outside the timed path: synthetic
    # conditioning tensors of the right shape are fed instead.
    for k in ("cond_stage_config", "img_cond_stage_config", "image_proj_stage_config"):
        p[k] = {"target": "torch.nn.Identity"}
    torch.manual_seed(1234)
    with torch.device(device):
        model = instantiate_from_config(cfg["model"])
    return model.to(device).eval(), cfg


def synth_inputs(res, device, seed):
    h, w = LATENT[res]
    g = torch.Generator(device="cpu").manual_seed(seed)
    T = 16
    cond_ctx = torch.randn(1, 77 + 16 * T, 1024, generator=g)
    uc_ctx = torch.randn(1, 77 + 16 * T, 1024, generator=g)
    first = torch.randn(1, 4, 1, h, w, generator=g) * 0.18215 * 4
    c_concat = first.repeat(1, 1, T, 1, 1)                       # image-to-video: frame-0 latent repeated
    return dict(cond_ctx=cond_ctx.to(device), uc_ctx=uc_ctx.to(device), c_concat=c_concat.to(device).contiguous())


def cpu_baseline(res_sample="256", threads=None):
    """Time the CPU oracle (oracle/unet.py, fp32) on a bounded sample: ONE UNet forward at the 256-config latent
    (16 frames x 32x32, same 1.44 B-parameter network), then scale by algorithmic FLOPs to the benchmarked
    workload. Reported, not a target."""
    from oracle import unet as ounet
    if threads is None:
        try:
            threads = len(os.sched_getaffinity(0))
        except AttributeError:
            threads = os.cpu_count() or 1
        threads = max(1, min(threads, 16))                   # the GPU box's CPU share for one GPU
    torch.set_num_threads(threads)
    cfg = ounet.UNetCfg(default_fs=10)
    shapes = ounet.unet_param_shapes(cfg)
    base = torch.randn(1 << 22)
    sd = {}
    for k, s in shapes.items():                                # cheap fill: timing does not depend on the values
        n = 1
        for d in s:
            n *= d
        reps = (n + base.numel() - 1) // base.numel()
        t = base.repeat(reps)[:n].reshape(s).clone() if n else torch.zeros(s)
        fan = max(1, n // max(1, s[0])) if len(s) > 1 else 1
        sd[k] = t * (0.5 / fan ** 0.5) if len(s) > 1 else (1.0 + 0.1 * t if k.endswith("weight") else 0.05 * t)
    h, w = LATENT[res_sample]
    x = torch.randn(1, 8, 16, h, w)
    ctx = torch.randn(1, 77 + 256, 1024)
    log(f"cpu_baseline: weights filled, running the forward on {threads} threads")
    t0 = time.perf_counter()
    ounet.unet_forward(sd, cfg, x, torch.tensor([500]), ctx, torch.tensor([10]))
    dt = time.perf_counter() - t0
    return dt, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", default="1024", choices=["256", "512", "1024"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ae", action="store_true")
    ap.add_argument("--no-trace", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    import torch.distributed as dist
    # DC_BENCH_BACKEND=gloo + DC_BENCH_SHARE_GPU=1 rehearse the N>1 code path on a one-GPU box (all ranks on cuda:0,
    # collectives on CPU tensors); the real launch uses RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("DC_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("DC_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    coll_dev = device if backend == "nccl" else torch.device("cpu")

    from dynamicrafter_amd import _hip, ops
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler, FusedRun
    _hip.lib()                                              # fail loudly if the HIP extension is missing

    res = args.res
    log(f"building inference_{res} model (random init) on {device}")
    model, cfg = build_model(res, device)
    log("model built")
    h, w = LATENT[res]
    T, S = 16, 50
    inp = synth_inputs(res, device, seed=7)
    if world > 1:                                            # conditioning broadcast once from rank 0 (RCCL/xGMI)
        for k in ("cond_ctx", "uc_ctx", "c_concat"):
            buf = inp[k].to(coll_dev)
            dist.broadcast(buf, src=0)
            inp[k] = buf.to(device)
    cond = {"c_crossattn": [inp["cond_ctx"]], "c_concat": [inp["c_concat"]]}
    uc = {"c_crossattn": [inp["uc_ctx"]], "c_concat": [inp["c_concat"]]}
    fs = torch.tensor([10], dtype=torch.long, device=device)
    shape = (1, 4, T, h, w)
    g = torch.Generator(device="cpu").manual_seed(100 + rank)      # every rank denoises its own clip
    x_T = torch.randn(shape, generator=g).to(device)
    noises = torch.randn((S,) + shape, generator=g).to(device)

    sampler = DDIMSampler(model)
    sampler.make_schedule(S, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    img = x_T.clone()
    run = FusedRun(sampler, img, [cond, uc], fs=fs, noises=noises, cfg_scale=7.5, guidance_rescale=0.7)
    log("capturing the step graph (1 eager warm-up step)")
    run.capture()
    log("graph captured")

    def do_steps(n, start):
        for i in range(n):
            if (start + i) % S == 0 and (start + i) > 0:
                run.rewind(x_T)                                   # next clip: counter -> 0, latent -> x_T
            run.step()

    do_steps(args.warmup, 0)
    run.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    l = _hip.lib()
    import ctypes as C
    e0, e1 = C.c_void_p(), C.c_void_p()
    l.dc_event_create(C.byref(e0)); l.dc_event_create(C.byref(e1))
    t0 = time.perf_counter()
    l.dc_event_record(e0, run.graph._stream)
    do_steps(args.steps, args.warmup)
    l.dc_event_record(e1, run.graph._stream)
    run.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ev_ms = C.c_float()
    l.dc_event_elapsed_ms(e0, e1, C.byref(ev_ms))
    if world > 1:
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    ms_per_step = elapsed / args.steps * 1e3
    log(f"timed {args.steps} steps: {ms_per_step:.2f} ms/step")
    finite = bool(torch.isfinite(run.img).all().item())

    # AutoencoderKL encode + decode of one 16-frame clip (per-frame, as perframe_ae=True)
    enc_ms = dec_ms = None
    if not args.no_ae:
        H, W = h * 8, w * 8
        video = torch.rand(1, 3, T, H, W, generator=torch.Generator().manual_seed(5)).mul(2).sub(1).to(device)
        z = model.encode_first_stage(video[:, :, :1])            # warm-up (allocations)
        model.decode_first_stage(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = model.encode_first_stage(video)
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - t0) * 1e3
        log(f"AE encode 16 frames: {enc_ms:.1f} ms")
        t0 = time.perf_counter()
        rec = model.decode_first_stage(z)
        torch.cuda.synchronize()
        dec_ms = (time.perf_counter() - t0) * 1e3
        log(f"AE decode 16 frames: {dec_ms:.1f} ms")
        finite = finite and bool(torch.isfinite(rec).all().item())
        del video, rec
    clip_s = S * ms_per_step / 1e3 + ((enc_ms or 0) + (dec_ms or 0)) / 1e3
    fps = world * T / clip_s
    guard = None
    if os.environ.get("DC_ARENA_GUARD", "0") == "1":
        # debugging aid: every scratch buffer sits between sentinel rows; raises if any launch wrote outside its buffer
        n = model.model.diffusion_model._arena.check() + model.first_stage_model._arena.check()
        guard = f"{n} scratch buffers verified"
        log(f"arena guard: {guard}")

    out = {
        "metric": "frames/sec (and denoising-step ms), 16f@576x1024, DDIM 50" if res == "1024" else f"frames/sec, 16f @{res} config",
        "value": round(fps, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (random-init weights, synthetic conditioning)",
        "config": {"workload": f"inference_{res}_v1.0.yaml: 1 clip/GPU, 16 frames, latent {h}x{w}, DDIM 50 "
                               "uniform_trailing eta=1, CFG 7.5 batched (cond+uncond), guidance_rescale 0.7, "
                               "v-param+ZTSNR+dynamic rescale, hipGraph-captured step",
                   "clips_per_gpu": 1, "parallelism": f"dp{world} over clips (no data-path collective)",
                   "ae_encode_ms": None if enc_ms is None else round(enc_ms, 1),
                   "ae_decode_ms": None if dec_ms is None else round(dec_ms, 1),
                   "clip_seconds": round(clip_s, 3), "outputs_finite": finite, "scratch_guard": guard,
                   "step_ms_hip_events": round(ev_ms.value / args.steps, 3)},
    }
    if world == 1 and A100_REF.get(res):
        out["config"]["reference_a100_s_per_clip_published"] = A100_REF[res]

    if rank == 0:
        step_tf = STEP_TFLOP[res]
        ach = step_tf / (ms_per_step / 1e3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"r01_hbm_traffic_step{res}.json")
        if os.path.exists(tpath):      # HBM bytes per step from rocprofv3 PMC passes (cannot be read live in-process)
            traffic = json.load(open(tpath)).get("hbm_bytes_per_step")
        roof = {"bound": "mfma", "kernel": "denoising step (hipGraph: 2 batched UNet forwards + DDIM update)",
                "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": "profiles/" + os.path.basename(tpath) if traffic else None,
                "algorithmic_tflop_per_launch": step_tf}
        if not args.no_trace:
            log("event-traced eager step")
            run.rewind(x_T)
            with ops.Tracer() as tr:
                run._enqueue()                                # one eager step on the current stream, event-bracketed
                torch.cuda.synchronize()
                fam = tr.summary()
            tot = sum(v["ms"] for v in fam.values())
            rows = []
            for name, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
                r = {"kernel": name, "launches": v["launches"], "ms": round(v["ms"], 3),
                     "share": round(v["ms"] / tot, 4), "avg_us": round(1e3 * v["ms"] / v["launches"], 2)}
                if v["flops"] > 0:
                    r["tflops"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)
                    r["frac_mfma"] = round(r["tflops"] / MFMA_PEAK_TFLOPS, 4)
                else:
                    r["gbs"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
                    r["frac_hbm"] = round(r["gbs"] / HBM_PEAK_GBS, 4)
                rows.append(r)
            roof["per_kernel_eager_step"] = rows
            if os.environ.get("DC_BENCH_DETAIL"):
                for (name, tag), v in sorted(tr.detail.items(), key=lambda kv: -kv[1]["ms"])[:40]:
                    log(f"  {name:22s} mode,M,N,K={tag}  x{v['launches']:3d}  {v['ms']:8.3f} ms  "
                        f"{v['flops'] / (v['ms'] * 1e-3) / 1e12:7.1f} TF/s")
            dom = rows[0]
            roof["dominant_kernel"] = {"name": dom["kernel"], "avg_launch_us": dom["avg_us"],
                                       "achieved": dom.get("tflops", dom.get("gbs")),
                                       "unit": "TFLOP/s" if "tflops" in dom else "GB/s",
                                       "frac": dom.get("frac_mfma", dom.get("frac_hbm"))}
        out["roofline"] = roof
        if not args.no_cpu_baseline:
            log("cpu_baseline: oracle UNet forward on host cores")
            dt, threads = cpu_baseline("256")
            log(f"cpu_baseline: {dt:.1f} s")
            scale = FWD_TFLOP[res] / FWD_TFLOP["256"]
            cpu_step_s = 2 * dt * scale
            out["cpu_baseline"] = {"value": round(T / (S * cpu_step_s), 6), "unit": "frames/s", "cores": threads,
                                   "kind": "port",
                                   "sample": f"oracle (CPU fp32 restatement) ONE UNet forward, 16 frames @32x32 latent "
                                             f"(4.9035 TFLOP) = {dt:.2f} s; scaled x{scale:.2f} by FLOPs to the "
                                             f"{h}x{w} latent, x2 forwards/step, x50 steps; AE excluded",
                                   "cpu_tflops": round(FWD_TFLOP['256'] / dt, 3),
                                   "step_seconds_scaled": round(cpu_step_s, 1)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


A100_REF = {"1024": 75.0, "512": 20.0, "256": 10.0}    # README.md:294-296 (end-to-end, A100), informational only

if __name__ == "__main__":
    main()
