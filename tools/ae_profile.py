"""Per-kernel-family time of AutoencoderKL encode / decode of ONE 576x1024 frame (HIP events around every launch)."""
import os, sys, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicrafter_amd import ops
from dynamicrafter_amd.utils.utils import instantiate_from_config
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "dynamicrafter_amd", "configs", "inference_1024_v1.0.yaml")))
ae = instantiate_from_config(cfg["model"]["params"]["first_stage_config"]).to("cuda:0")
NF = int(os.environ.get("AE_FRAMES", "1"))          # frames per call (production: ae_frames_per_call = 4)
x = torch.rand(NF, 3, 576, 1024, device="cuda:0") * 2 - 1
z = ae.encode(x).mode(); ae.decode(z); torch.cuda.synchronize()
for name, fn in (("encode", lambda: ae.encode(x).mode()), ("decode", lambda: ae.decode(z))):
    with ops.Tracer() as tr:
        fn(); torch.cuda.synchronize()
        fam = tr.summary()
    tot = sum(v["ms"] for v in fam.values())
    print(f"== {name}: {tot:.2f} ms per call of {NF} frame(s) = {tot / NF:.2f} ms per frame")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        extra = f"{v['flops'] / (v['ms'] * 1e-3) / 1e12:7.1f} TF/s" if v["flops"] > 0 else f"{v['bytes'] / (v['ms'] * 1e-3) / 1e9:7.1f} GB/s"
        print(f"   {k:42s} x{v['launches']:3d} {v['ms']:8.3f} ms  {extra}")
    if os.environ.get("AE_DETAIL"):
        for (n, tag), v in sorted(tr.detail.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("AE_DETAIL_N", "14"))]:
            print(f"      {n:40s} {tag} x{v['launches']} {v['ms']:.3f} ms {v['flops'] / (v['ms'] * 1e-3) / 1e12:.0f} TF/s")

# ---- frames per launch: GroupNorm / attention are per frame, so batching frames only changes how many rows a launch sees
if os.environ.get("AE_BATCH"):
    import time
    xs = torch.rand(16, 3, 576, 1024, device="cuda:0") * 2 - 1
    for nb in (1, 2, 4, 8, 16):
        def enc():
            return torch.cat([ae.encode(xs[i:i + nb]).mode() for i in range(0, 16, nb)])
        zz = enc(); torch.cuda.synchronize()
        t0 = time.perf_counter(); zz = enc(); torch.cuda.synchronize(); te = (time.perf_counter() - t0) * 1e3
        def dec():
            return torch.cat([ae.decode(zz[i:i + nb]) for i in range(0, 16, nb)])
        rr = dec(); torch.cuda.synchronize()
        t0 = time.perf_counter(); rr = dec(); torch.cuda.synchronize(); td = (time.perf_counter() - t0) * 1e3
        if nb == 1:
            z1, r1 = zz.clone(), rr.clone()
        print(f"frames/launch {nb:2d}: encode {te:7.1f} ms  decode {td:7.1f} ms per 16-frame clip | "
              f"bitwise == per-frame: enc {bool(torch.equal(zz, z1))} dec {bool(torch.equal(rr, r1))} "
              f"(max abs diff {float((zz - z1).abs().max()):.2e} / {float((rr - r1).abs().max()):.2e})", flush=True)
