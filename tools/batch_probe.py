"""Is the batched (cond+uncond) UNet forward faster or slower than two single-branch forwards? (cache locality of
producer -> consumer tensors vs weight reuse / launch efficiency). Prints ms per forward for B_eff = 1 and 2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler, FusedRun
dev = torch.device("cuda", 0)
res = sys.argv[1] if len(sys.argv) > 1 else "1024"
model, cfg = bench.build_model(res, dev)
h, w = bench.LATENT[res]
inp = bench.synth_inputs(res, dev, seed=7)
cond = {"c_crossattn": [inp["cond_ctx"]], "c_concat": [inp["c_concat"]]}
uc = {"c_crossattn": [inp["uc_ctx"]], "c_concat": [inp["c_concat"]]}
fs = torch.tensor([10], dtype=torch.long, device=dev)
shape = (1, 4, 16, h, w)
x_T = torch.randn(shape, device=dev); noises = torch.randn((50,) + shape, device=dev)
for name, branches, scale in (("2 branches batched", [cond, uc], 7.5), ("1 branch", [cond], 1.0)):
    for share in ("1", "0"):
        os.environ["DC_SHARED_PREFIX"] = share
        s = DDIMSampler(model); s.make_schedule(50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
        run = FusedRun(s, x_T.clone(), branches, fs=fs, noises=noises, cfg_scale=scale, guidance_rescale=0.7 if len(branches) > 1 else 0.0)
        run.capture()
        for _ in range(2): run.step()
        run.sync(); torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(6): run.step()
        run.sync(); torch.cuda.synchronize()
        print(f"{name:20s} shared_prefix={share}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms/step", flush=True)
        if len(branches) == 1: break
