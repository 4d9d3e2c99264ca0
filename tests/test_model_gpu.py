"""Model-level parity on the GPU: the HIP-backed lvdm classes against (a) golden vectors captured from the
reference (tests/golden) and (b) the CPU oracle on the same seeded inputs.

Stated tolerances (bf16 storage, fp32 accumulate; oracle/reference are fp32), each <= 2x the value measured on MI355X
(every test prints what it measures):
  whole UNet forward      rel-L2 <= 3e-2 and cosine >= 0.999          (measured 1.7e-2 / 0.99986)
  AutoencoderKL enc/dec   rel-L2 <= 3e-2                               (measured 1.5e-2 moments, 1.7e-2 decode)
  10-step DDIM trajectory rel-L2 <= 8.5e-2                             (measured 3.1e-2 / 4.2e-2)
  50-step trajectory      rel-L2 <= 4e-2                               (measured 2.0e-2)
  fused DDIM step (fp32)  max-rel <= 1e-5
"""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def T(a, dev=DEV):
    return torch.from_numpy(np.asarray(a)).to(dev)


def rel_l2(a, b):
    a = torch.as_tensor(a).float().cpu(); b = torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def cosine(a, b):
    a = torch.as_tensor(a).float().cpu().flatten(); b = torch.as_tensor(b).float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm())).item()


def maxrel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def recipe_load(module, seed):
    from oracle.weights import fill_state_dict
    sd = module.state_dict()
    module.load_state_dict(fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed), strict=True)
    return module


@pytest.mark.parametrize("tag", ["v1024", "v256"])
def test_unet_tiny_vs_reference(tag):
    from dynamicrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    g = load(f"unet_tiny_{tag}")
    params = yaml.safe_load(str(g["yaml_params"]))
    net = UNetModel(**params)
    assert sorted(net.state_dict().keys()) == [str(s) for s in g["param_names"]]
    recipe_load(net, 11).to(DEV)
    y = net(T(g["x"]), T(g["timesteps"]), context=T(g["context"]), fs=T(g["fs"]))
    assert y.shape == g["y"].shape and y.dtype == torch.float32
    print(f"\n[unet tiny {tag}] vs reference rel-L2 {rel_l2(y, g['y']):.3e} cosine {cosine(y, g['y']):.6f}")
    assert rel_l2(y, g["y"]) < 3e-2 and cosine(y, g["y"]) > 0.999
    y2 = net(T(g["x"]), T(g["timesteps"]), context=T(g["context"]))
    assert rel_l2(y2, g["y_default_fs"]) < 3e-2
    # determinism: identical launches give identical bits
    y3 = net(T(g["x"]), T(g["timesteps"]), context=T(g["context"]))
    assert torch.equal(y2, y3)


def test_unet_fullwidth_vs_reference():
    """Released architecture (1.44 B params) at an 8x8 latent, 16 frames; weights from the seeded recipe."""
    from dynamicrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    g = load("unet_fullwidth_8x8")
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(G), "..", "dynamicrafter_amd", "configs",
                                           "inference_1024_v1.0.yaml")))
    params = cfg["model"]["params"]["unet_config"]["params"]
    net = UNetModel(**params)
    digest = sorted(f"{k}:{'x'.join(map(str, v.shape))}" for k, v in net.state_dict().items())
    assert digest == [str(s) for s in g["key_digest"]]
    recipe_load(net, 12).to(DEV)
    y = net(T(g["x"]), T(g["timesteps"]), context=T(g["context"]), fs=T(g["fs"]))
    print(f"\n[unet fullwidth 8x8] vs reference rel-L2 {rel_l2(y, g['y']):.3e} cosine {cosine(y, g['y']):.6f}")
    assert rel_l2(y, g["y"]) < 3e-2 and cosine(y, g["y"]) > 0.999


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_autoencoder_vs_reference(tag):
    from dynamicrafter_amd.lvdm.models.autoencoder import AutoencoderKL
    g = load(f"ae_{tag}")
    dd = yaml.safe_load(str(g["yaml_params"]))
    ae = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)
    assert sorted(ae.state_dict().keys()) == [str(s) for s in g["param_names"]]
    recipe_load(ae, 13).to(DEV)
    post = ae.encode(T(g["img"]))
    assert rel_l2(post.parameters, g["moments"]) < 3e-2
    z = post.sample(noise=T(g["noise"]))
    assert rel_l2(z, g["z"]) < 3e-2
    assert rel_l2(post.mode(), g["z_mode"]) < 3e-2
    rec = ae.decode(T(g["z"]))
    assert rec.shape == g["rec"].shape
    print(f"\n[ae {tag}] moments {rel_l2(post.parameters, g['moments']):.3e} z {rel_l2(z, g['z']):.3e} "
          f"decode {rel_l2(rec, g['rec']):.3e}")
    assert rel_l2(rec, g["rec"]) < 3e-2


def _tiny_lvd(config_name, unet_extra=None, seed_unet=11, seed_ae=13):
    """LatentVisualDiffusion from this package's YAML with the golden generator's tiny nets."""
    from dynamicrafter_amd.utils.utils import instantiate_from_config
    from tests.golden_cfg import TINY_AE, TINY_UNET
    root = os.path.join(os.path.dirname(G), "..", "dynamicrafter_amd", "configs")
    cfg = yaml.safe_load(open(os.path.join(root, config_name)))
    p = cfg["model"]["params"]
    p["unet_config"]["params"] = dict(TINY_UNET, default_fs=p["unet_config"]["params"]["default_fs"], **(unet_extra or {}))
    p["first_stage_config"]["params"]["ddconfig"] = dict(TINY_AE)
    for k in ("cond_stage_config", "img_cond_stage_config", "image_proj_stage_config"):
        p[k] = {"target": "torch.nn.Identity"}
    model = instantiate_from_config(cfg["model"])
    recipe_load(model.model.diffusion_model, seed_unet)
    recipe_load(model.first_stage_model, seed_ae)
    return model.to(DEV)


class _Fake:
    """Schedule carrier whose apply_model returns queued tensors (generic sampler path)."""
    def __init__(self, real, outs):
        self.__dict__.update({k: getattr(real, k) for k in ("num_timesteps", "alphas_cumprod", "betas",
                              "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                              "parameterization", "use_dynamic_rescale")})
        if real.use_dynamic_rescale:
            self.scale_arr = real.scale_arr
        self.device = torch.device(DEV)
        self._outs = list(outs)

    def apply_model(self, x, t, c, **kw):
        return self._outs.pop(0)


def test_p_sample_ddim_known_answers():
    """The fused fp32 DDIM kernel against the reference's p_sample_ddim outputs (2- and 3-branch CFG)."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    g = load("p_sample_ddim")
    x, ec, eu, ei, noise = (T(g[k]) for k in ("x", "e_cond", "e_uncond", "e_img", "noise"))
    n = 0
    for cname, disc, gr in (("inference_256_v1.0.yaml", "uniform", 0.0), ("inference_512_v1.0.yaml", "uniform_trailing", 0.7),
                            ("inference_1024_v1.0.yaml", "uniform_trailing", 0.7)):
        model = _tiny_lvd(cname)
        tag = cname.split("_")[1]
        for eta in (0, 1):
            for index in (9, 4, 0):
                for nm in ("cfg2", "cfg3"):
                    key = f"{tag}/{disc}/eta{eta}/i{index}/{nm}"
                    s = DDIMSampler(_Fake(model, [ec, eu] + ([ei] if nm == "cfg3" else [])))
                    s.make_schedule(10, ddim_discretize=disc, ddim_eta=float(eta), verbose=False)
                    ts = torch.full((x.shape[0],), 1, device=DEV, dtype=torch.long)
                    extra = dict(cfg_img=2.0, unconditional_conditioning_img_nonetext={"c": 2}) if nm == "cfg3" else {}
                    xp, px0 = s.p_sample_ddim(x.clone(), {"c": 1}, ts, index=index, unconditional_guidance_scale=7.5,
                                              unconditional_conditioning={"c": 0}, guidance_rescale=gr, noise=noise,
                                              **extra)
                    if not np.isfinite(g[key + "/x_prev"]).all():
                        assert torch.isfinite(xp).all()
                        continue
                    assert maxrel(px0, g[key + "/pred_x0"]) < 1e-5, key
                    assert maxrel(xp, g[key + "/x_prev"]) < 1e-5, key
                    n += 1
    assert n >= 30


@pytest.mark.parametrize("tag,disc,eta,gr,extra", [
    ("256", "uniform", 0.0, 0.0, dict(image_cross_attention_scale_learnable=True)),
    ("512", "uniform_trailing", 1.0, 0.7, dict())])
def test_ddim_trajectory_vs_reference(tag, disc, eta, gr, extra):
    """10 DDIM steps with CFG 7.5 through sampler + hybrid conditioning + UNet, eager and hipGraph-replayed."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    g = load(f"trajectory_{tag}")
    model = _tiny_lvd(f"inference_{tag}_v1.0.yaml", extra)
    cond = {"c_crossattn": [T(g["ctx"])], "c_concat": [T(g["c_concat"])]}
    uc = {"c_crossattn": [T(g["uc_ctx"])], "c_concat": [T(g["c_concat"])]}
    x_T = T(g["x_T"])
    outs = []
    for use_graph in (False, True):
        s = DDIMSampler(model)
        samples, inter = s.sample(S=10, batch_size=1, shape=tuple(x_T.shape[1:]), conditioning=cond, verbose=False,
                                  unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=eta, x_T=x_T,
                                  fs=T(g["fs"]), timestep_spacing=disc, guidance_rescale=gr,
                                  noises=T(g["noises"]) if eta > 0 else None, use_graph=use_graph)
        assert torch.isfinite(samples).all()
        r = rel_l2(samples, g["samples"])
        if not use_graph:
            print(f"\n[trajectory {tag}] 10 steps vs reference rel-L2 {r:.3e}")
        assert r < 8.5e-2                                  # measured 3.1e-2 (256 cfg) / 4.2e-2 (512 cfg, eta=1)
        outs.append(samples.clone())
    assert torch.equal(outs[0], outs[1])          # graph replay == eager launches, bit for bit
    assert torch.equal(x_T, T(g["x_T"]))          # inputs are not mutated


def test_first_stage_vs_reference():
    g = load("first_stage")
    model = _tiny_lvd("inference_512_v1.0.yaml")
    vid = T(g["video"])
    noise = torch.from_numpy(g["noise"])
    it = iter([noise[i:i + 1] for i in range(noise.shape[0])])
    orig = torch.randn
    torch.randn = lambda *a, **k: next(it)            # posterior.sample() draws CPU torch.randn(shape), as the reference
    try:
        z = model.encode_first_stage(vid)
    finally:
        torch.randn = orig
    assert z.shape == g["z"].shape
    assert rel_l2(z, g["z"]) < 3e-2
    rec = model.decode_first_stage(T(g["z"]))
    assert rel_l2(rec, g["rec"]) < 3e-2


def test_unet_vs_oracle_odd_sizes():
    """Oracle comparison on a latent whose row counts are not multiples of the 128-row GEMM tile, T=3, B=1,
    and a context without per-frame image tokens... exercises tail masking in every kernel."""
    from dynamicrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    from oracle import unet as ounet
    from oracle.weights import fill_state_dict
    from tests.golden_cfg import TINY_UNET
    params = dict(TINY_UNET, temporal_length=3)
    cfg = ounet.UNetCfg.from_params(params)
    sd = fill_state_dict(ounet.unet_param_shapes(cfg), seed=5)
    net = UNetModel(**params)
    net.load_state_dict(sd, strict=True)
    net.to(DEV)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 8, 3, 24, 8, generator=gen)
    ctx = torch.randn(1, 77 + 3 * 16, 128, generator=gen)
    ts = torch.tensor([333])
    ref = ounet.unet_forward(sd, cfg, x, ts, ctx, None)
    y = net(x.to(DEV), ts.to(DEV), context=ctx.to(DEV))
    assert rel_l2(y, ref) < 3e-2 and cosine(y, ref) > 0.999


def test_no_cpu_fallback():
    from dynamicrafter_amd.lvdm.modules.networks.openaimodel3d import UNetModel
    from tests.golden_cfg import TINY_UNET
    net = UNetModel(**TINY_UNET)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 8, 4, 8, 8), torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 141, 128))


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_resampler_vs_reference(tag):
    """the image-token projector on the same GEMM / LayerNorm / flash-attention kernels."""
    from dynamicrafter_amd.lvdm.modules.encoders.resampler import Resampler
    g = load(f"resampler_{tag}")
    kw = yaml.safe_load(str(g["yaml_params"]))
    m = Resampler(**kw)
    assert sorted(m.state_dict().keys()) == [str(s) for s in g["param_names"]]
    recipe_load(m, 14).to(DEV)
    y = m(T(g["x"]))
    assert y.shape == g["y"].shape and y.dtype == torch.float32
    assert rel_l2(y, g["y"]) < 3e-2 and cosine(y, g["y"]) > 0.999


def test_three_branch_cfg_fused_vs_oracle():
    """cfg_img / image-only unconditional branch (ddim_multiplecond.py) through the fused batch-3 path, 4 steps."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    from oracle import ddim as oddim
    from oracle import unet as ounet
    from oracle.weights import fill_state_dict
    from tests.golden_cfg import TINY_UNET
    model = _tiny_lvd("inference_512_v1.0.yaml")
    params = dict(TINY_UNET, default_fs=24)
    cfg = ounet.UNetCfg.from_params(params)
    sd = fill_state_dict(ounet.unet_param_shapes(cfg), seed=11)
    g = torch.Generator().manual_seed(9)
    b, t, h, w = 1, 4, 16, 16
    x_T = torch.randn(b, 4, t, h, w, generator=g)
    ctx = [torch.randn(b, 77 + 16 * t, 128, generator=g) for _ in range(3)]
    cc = torch.randn(b, 4, t, h, w, generator=g) * 0.2
    noises = torch.randn(4, b, 4, t, h, w, generator=g)
    fs = torch.tensor([24])
    ms = oddim.ModelSchedule(rescale_betas_zero_snr=True, parameterization="v", use_dynamic_rescale=True, base_scale=0.7)
    sc = oddim.DDIMSchedule(ms, 4, "uniform_trailing", 1.0)
    ref = oddim.ddim_sample(lambda x, tl, c, fs=None: ounet.unet_forward(sd, cfg, torch.cat([x, cc], 1), tl, c, fs), sc,
                            x_T, ctx[0], ctx[1], cfg_scale=7.5, guidance_rescale=0.7, noises=list(noises),
                            uncond_img=ctx[2], cfg_img=2.0, fs=fs)
    mk = lambda c: {"c_crossattn": [c.to(DEV)], "c_concat": [cc.to(DEV)]}
    s = DDIMSampler(model)
    out, _ = s.sample(S=4, batch_size=b, shape=(4, t, h, w), conditioning=mk(ctx[0]), verbose=False,
                      unconditional_guidance_scale=7.5, unconditional_conditioning=mk(ctx[1]), eta=1.0, x_T=x_T.to(DEV),
                      fs=fs.to(DEV), timestep_spacing="uniform_trailing", guidance_rescale=0.7, noises=noises.to(DEV),
                      cfg_img=2.0, unconditional_conditioning_img_nonetext=mk(ctx[2]))
    assert rel_l2(out, ref) < 1e-1
    # generic (non-fused) path of the same sampler gives the same trajectory within bf16 noise
    class Plain:                                  # hides apply_model_rows -> forces separate apply_model calls
        def __init__(self, m): self._m = m
        def __getattr__(self, k):
            if k in ("apply_model_rows", "prepare_branches"): raise AttributeError(k)
            return getattr(self._m, k)
    out2, _ = DDIMSampler(Plain(model)).sample(S=4, batch_size=b, shape=(4, t, h, w), conditioning=mk(ctx[0]),
                                               verbose=False, unconditional_guidance_scale=7.5,
                                               unconditional_conditioning=mk(ctx[1]), eta=1.0, x_T=x_T.to(DEV),
                                               fs=fs.to(DEV), timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                               noises=noises.to(DEV), cfg_img=2.0,
                                               unconditional_conditioning_img_nonetext=mk(ctx[2]))
    assert rel_l2(out2, ref) < 1e-1 and rel_l2(out2, out) < 5e-2


def test_shared_prefix_is_bit_identical(monkeypatch):
    """cond/uncond share latent + c_concat: computing the context-free prefix once must not change a single bit."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    g = load("trajectory_512")
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("DC_SHARED_PREFIX", flag)
        model = _tiny_lvd("inference_512_v1.0.yaml")
        cc = T(g["c_concat"])
        cond = {"c_crossattn": [T(g["ctx"])], "c_concat": [cc]}
        uc = {"c_crossattn": [T(g["uc_ctx"])], "c_concat": [cc]}
        s = DDIMSampler(model)
        x_T = T(g["x_T"])
        samples, _ = s.sample(S=3, batch_size=1, shape=tuple(x_T.shape[1:]), conditioning=cond, verbose=False,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0, x_T=x_T,
                              fs=T(g["fs"]), timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                              noises=T(g["noises"])[:3])
        assert s._last_run.prep["share"] == (2 if flag == "1" else 1)
        outs.append(samples.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("tag,cname", [("a", "inference_512_v1.0.yaml"), ("b", "inference_256_v1.0.yaml"),
                                       ("c", "inference_512_v1.0.yaml")])
def test_image_guided_synthesis_vs_reference(tag, cname):
    """The host harness end to end: toy CLIP stand-ins -> HIP Resampler -> conditioning assembly
    (c_concat repeat / interp first+last, uc, third branch) -> DDIM loop -> per-frame AE decode, against the output of
    the reference's scripts/evaluation/inference.py:image_guided_synthesis on the same injected tensors.
    Tolerance: the 10-step trajectory bound (rel-L2 <= 1e-1) carried through the AE decoder."""
    from dynamicrafter_amd.scripts.evaluation.inference import image_guided_synthesis
    from dynamicrafter_amd.utils.utils import instantiate_from_config
    from tests.golden_cfg import TINY_AE, TINY_RESAMPLER, TINY_UNET
    g = load(f"harness_{tag}")
    kw = yaml.safe_load(str(g["kwargs"]))
    root = os.path.join(os.path.dirname(G), "..", "dynamicrafter_amd", "configs")
    cfg = yaml.safe_load(open(os.path.join(root, cname)))
    p = cfg["model"]["params"]
    extra = dict(image_cross_attention_scale_learnable=True) if "256" in cname else {}
    p["unet_config"]["params"] = dict(TINY_UNET, default_fs=p["unet_config"]["params"]["default_fs"], **extra)
    p["first_stage_config"]["params"]["ddconfig"] = dict(TINY_AE)
    p["cond_stage_config"] = {"target": "tests.golden_cfg.ToyTextEmbedder"}
    p["img_cond_stage_config"] = {"target": "tests.golden_cfg.ToyImageEmbedder"}
    p["image_proj_stage_config"] = {"target": "lvdm.modules.encoders.resampler.Resampler", "params": dict(TINY_RESAMPLER)}
    model = instantiate_from_config(cfg["model"])
    recipe_load(model.model.diffusion_model, 11)
    recipe_load(model.first_stage_model, 13)
    recipe_load(model.image_proj_model, 14)
    model = model.to(DEV)
    videos = T(g["videos"])
    b, _, t, H, W = videos.shape
    ae_noise = torch.from_numpy(g["ae_noise"])
    it = iter([ae_noise[i:i + 1] for i in range(ae_noise.shape[0])])
    orig = torch.randn
    torch.randn = lambda *a, **k: next(it)            # posterior.sample() draws CPU torch.randn(shape), as the reference
    try:
        prompt = "two frames of a blooming flower" if tag == "c" else "a corgi running on the beach"
        out = image_guided_synthesis(model, [prompt], videos, [b, 4, t, H // 8, W // 8],
                                     n_samples=1, x_T=T(g["x_T"]), noises=T(g["noises"]), **kw)
    finally:
        torch.randn = orig
    assert tuple(out.shape) == tuple(g["out"].shape)
    assert torch.isfinite(out).all()
    r = rel_l2(out, g["out"])
    print(f"\n[harness {tag}] decoded clip vs reference rel-L2 {r:.3e}")
    assert r < {"a": 1e-1, "b": 3e-2, "c": 1e-1}[tag]       # measured 6.3e-2 / 1.4e-2 / 7.3e-2
    assert torch.equal(videos, T(g["videos"]))       # inputs are not mutated


@pytest.mark.parametrize("hw", [8, 16])
def test_no_scratch_buffer_overrun(monkeypatch, hw):
    """Every scratch buffer of the UNet / AE / Resampler is allocated between sentinel rows (DC_ARENA_GUARD=1) and the
    sentinels must survive 2- and 3-branch sampling (shared guidance prefix on), encode, decode and the projector.
    Regression for the shared-prefix embedding overrun (nb rows written into a 1-row buffer)."""
    monkeypatch.setenv("DC_ARENA_GUARD", "1")
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    from dynamicrafter_amd.lvdm.modules.encoders.resampler import Resampler
    from tests.golden_cfg import TINY_RESAMPLER
    model = _tiny_lvd("inference_256_v1.0.yaml", dict(image_cross_attention_scale_learnable=True))
    g = torch.Generator().manual_seed(3)
    b, t = 1, 4
    mk = lambda: {"c_crossattn": [torch.randn(b, 77 + 16 * t, 128, generator=g).to(DEV)],
                  "c_concat": [(torch.randn(b, 4, t, hw, hw, generator=g) * 0.2).to(DEV)]}
    cond, uc, uc2 = mk(), mk(), mk()
    uc["c_concat"] = uc2["c_concat"] = cond["c_concat"]                 # shared prefix applies
    for extra in (dict(cfg_img=2.0, unconditional_conditioning_img_nonetext=uc2), {}):
        out, _ = DDIMSampler(model).sample(S=2, batch_size=b, shape=(4, t, hw, hw), conditioning=cond, verbose=False,
                                           unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0,
                                           x_T=torch.randn(b, 4, t, hw, hw, generator=g).to(DEV), **extra)
        assert torch.isfinite(out).all()
    vid = torch.randn(1, 3, t, 8 * hw, 8 * hw, generator=g).clamp(-1, 1).to(DEV)
    z = model.encode_first_stage(vid)
    rec = model.decode_first_stage(z)
    assert torch.isfinite(rec).all()
    rs = recipe_load(Resampler(**TINY_RESAMPLER), 14).to(DEV)
    y = rs(torch.randn(2, 9, 64, generator=g).to(DEV))
    assert torch.isfinite(y).all()
    n = model.model.diffusion_model._arena.check() + model.first_stage_model._arena.check() + rs._arena.check()
    assert n > 50


def test_ddim_trajectory_50_steps_vs_reference():
    """The length the bench runs: 50 eta=1 steps (v-param + ZTSNR + dynamic rescale + guidance rescale, CFG 7.5,
    uniform_trailing) of the tiny 512-config model against the reference's own 50-step run, eager and graph-replayed.
    Measures the drift of the bf16 activation path over a full-length trajectory."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    g = load("trajectory50_512")
    model = _tiny_lvd("inference_512_v1.0.yaml")
    cond = {"c_crossattn": [T(g["ctx"])], "c_concat": [T(g["c_concat"])]}
    uc = {"c_crossattn": [T(g["uc_ctx"])], "c_concat": [T(g["c_concat"])]}
    x_T = T(g["x_T"])
    outs = []
    for use_graph in (False, True):
        s = DDIMSampler(model)
        samples, inter = s.sample(S=50, batch_size=1, shape=tuple(x_T.shape[1:]), conditioning=cond, verbose=False,
                                  unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0, x_T=x_T,
                                  fs=T(g["fs"]), timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                  noises=T(g["noises"]), use_graph=use_graph, log_every_t=10)
        outs.append(samples.clone())
        if not use_graph:
            ref_i = g["x_inter"]
            drift = [rel_l2(a, b) for a, b in zip(inter["x_inter"][1:], ref_i[1:])]
            print("\n[trajectory50] rel-L2 after steps 1,10,20,30,40,50: " + " ".join(f"{d:.2e}" for d in drift))
    r = rel_l2(outs[0], g["samples"])
    print(f"[trajectory50] final sample vs reference rel-L2 {r:.3e}")
    assert r < 4e-2                                        # measured 2.0e-2 (drift grows ~linearly: 2e-3 per 5 steps)
    assert torch.equal(outs[0], outs[1])


def test_sampler_mask_decode_stochastic_encode_vs_reference():
    """SURVEY 8(f) rank 2 remainder on the HIP path against the reference sampler's outputs: mask / x0 blending (with
    and without clean_cond; ddim.py:174-180), decode (:281-301), stochastic_encode (:303-317)."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    g = load("sampler_extras")
    model = _tiny_lvd("inference_512_v1.0.yaml")
    cond = {"c_crossattn": [T(g["ctx"])], "c_concat": [T(g["c_concat"])]}
    uc = {"c_crossattn": [T(g["uc_ctx"])], "c_concat": [T(g["c_concat"])]}
    x_T, x0, mask = T(g["x_T"]), T(g["x0"]), T(g["mask"])
    for tag, clean in (("mask", False), ("mask_clean", True)):
        s = DDIMSampler(model)
        out, _ = s.sample(S=6, batch_size=1, shape=tuple(x_T.shape[1:]), conditioning=cond, verbose=False, mask=mask, x0=x0,
                          unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0, x_T=x_T, fs=T(g["fs"]),
                          timestep_spacing="uniform_trailing", guidance_rescale=0.7, clean_cond=clean,
                          noises=T(g["noises"]), q_noises=T(g["qnoises"]))
        r = rel_l2(out, g[f"{tag}/samples"])
        print(f"\n[sampler extras] {tag}: rel-L2 {r:.3e}")
        assert r < 7.5e-2                                  # measured 3.7e-2 / 3.4e-2 (6 steps, CFG 7.5, tiny net)
        assert torch.equal(out * mask, out * mask) and torch.isfinite(out).all()
    s = DDIMSampler(model)
    s.make_schedule(6, ddim_discretize="uniform", ddim_eta=0.0, verbose=False)
    dec = s.decode(T(g["decode/x_latent"]), cond, int(g["decode/t_start"]), unconditional_guidance_scale=7.5,
                   unconditional_conditioning=uc)
    r = rel_l2(dec, g["decode/x_dec"])
    print(f"[sampler extras] decode: rel-L2 {r:.3e}")
    assert r < 1e-1                                        # measured 5.0e-2
    n = T(g["enc/noise"])
    assert maxrel(s.stochastic_encode(x0, T(g["enc/t"]), noise=n), g["enc/ddim"]) < 1e-6
    assert maxrel(s.stochastic_encode(x0, T(g["enc/t_orig"]), use_original_steps=True, noise=n), g["enc/orig"]) < 1e-6


def test_sampler_mask_default_noise_draw_order_and_graph_bookkeeping():
    """With a mask and no injected noises the reference draws, per step, q_sample's randn_like and then the step's noise
    (ddim.py:174-180, :270): the fused path pre-draws both in that interleaved order, so the same seed gives the same sample
    as drawing them step by step and injecting them. Under use_graph=True callback / img_callback / intermediates behave as
    in the eager loop (the step has finished when they run)."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler
    g = load("sampler_extras")
    model = _tiny_lvd("inference_512_v1.0.yaml")
    cond = {"c_crossattn": [T(g["ctx"])], "c_concat": [T(g["c_concat"])]}
    uc = {"c_crossattn": [T(g["uc_ctx"])], "c_concat": [T(g["c_concat"])]}
    x_T, x0, mask = T(g["x_T"]), T(g["x0"]), T(g["mask"])
    S = 6
    kw = dict(S=S, batch_size=1, shape=tuple(x_T.shape[1:]), conditioning=cond, verbose=False, mask=mask, x0=x0,
              unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0, x_T=x_T, fs=T(g["fs"]),
              timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    torch.manual_seed(4321)
    a, _ = DDIMSampler(model).sample(**kw)
    torch.manual_seed(4321)                                   # the reference's order, drawn by hand and injected
    qs, ns = [], []
    for _ in range(S):
        qs.append(torch.randn(x_T.shape, device=DEV)); ns.append(torch.randn(x_T.shape, device=DEV))
    b, _ = DDIMSampler(model).sample(noises=torch.stack(ns), q_noises=torch.stack(qs), **kw)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):                           # a noise buffer shorter than S steps is refused up front
        DDIMSampler(model).sample(noises=torch.stack(ns[:3]), q_noises=torch.stack(qs), **kw)
    # eta = 0 with a mask: the reference still calls noise_like every step (ddim.py:270; sigma_t = 0 discards the value), so
    # q_sample's draws stay interleaved with one discarded draw per step - same seed, same q noises as the hand-drawn order
    kw0 = dict(kw, eta=0.0)
    torch.manual_seed(777)
    a0, _ = DDIMSampler(model).sample(**kw0)
    torch.manual_seed(777)
    qs0 = []
    for _ in range(S):
        qs0.append(torch.randn(x_T.shape, device=DEV)); torch.randn(x_T.shape, device=DEV)
    b0, _ = DDIMSampler(model).sample(q_noises=torch.stack(qs0), **kw0)
    assert torch.equal(a0, b0)
    torch.manual_seed(777)                                    # without a mask the generator advances by S draws as well
    DDIMSampler(model).sample(**dict(kw0, mask=None, x0=None))
    after = torch.randn(4, device=DEV)
    torch.manual_seed(777)
    for _ in range(S):
        torch.randn(x_T.shape, device=DEV)
    assert torch.equal(after, torch.randn(4, device=DEV))
    # bookkeeping under the captured graph == eager
    rec = {}
    for use_graph in (False, True):
        calls, snaps = [], []
        torch.manual_seed(99)
        out, inter = DDIMSampler(model).sample(callback=lambda i: calls.append(i),
                                               img_callback=lambda p, i: snaps.append((i, p.clone())), log_every_t=2,
                                               use_graph=use_graph, **kw)
        rec[use_graph] = (out, inter, calls, snaps)
    (o0, i0, c0, s0), (o1, i1, c1, s1) = rec[False], rec[True]
    assert torch.equal(o0, o1) and c0 == c1 == list(range(S)) and len(s0) == len(s1) == S
    assert all(torch.equal(p0, p1) for (_, p0), (_, p1) in zip(s0, s1))
    assert len(i0["x_inter"]) == len(i1["x_inter"]) > 2
    assert all(torch.equal(x, y) for x, y in zip(i0["x_inter"], i1["x_inter"]))
    assert all(torch.equal(x, y) for x, y in zip(i0["pred_x0"], i1["pred_x0"]))


def test_frames_to_uint8_and_writers(tmp_path):
    """Output side (inference.py:115-162): clamp / (v+1)/2 / x255 / uint8 truncation / side-by-side grid in one kernel,
    bit-exact against the reference's torch expression; APNG container round trip."""
    from dynamicrafter_amd.utils.save_video import frames_to_uint8, save_results, save_results_seperate
    from tests.test_host_cpu import _png_decode
    g = torch.Generator().manual_seed(12)
    video = torch.randn(3, 3, 5, 24, 40, generator=g) * 0.8              # values beyond [-1, 1] exercise the clamp
    video[0, 0, 0, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, float(np.nextafter(np.float32(1.0), np.float32(0)))])
    v = torch.clamp(video.float(), -1., 1.).permute(2, 0, 1, 3, 4)        # t n c h w
    grid = torch.stack([torch.cat(list(fr), dim=2) for fr in v])          # make_grid(nrow=n, padding=0): [t, c, h, n*w]
    ref = (((grid + 1.0) / 2.0) * 255).to(torch.uint8).permute(0, 2, 3, 1)
    out = frames_to_uint8(video.to(DEV))
    assert out.dtype == torch.uint8 and tuple(out.shape) == (5, 24, 120, 3)
    assert torch.equal(out.cpu(), ref)
    p = save_results("a prompt", video.to(DEV), "clip0001.mp4", str(tmp_path / "samples"), fps=8)
    dec, fps = _png_decode(p)
    assert p.endswith("clip0001.png") and fps == 8 and np.array_equal(dec, ref.numpy())
    ps = save_results_seperate("a prompt", video.to(DEV), "clip0001.mp4", str(tmp_path / "samples"), fps=10, loop=True)
    assert len(ps) == 3 and "samples_separate" in ps[0]
    dec, _ = _png_decode(ps[1])
    assert np.array_equal(dec, ref.numpy()[:-1, :, 40:80])                # loop: last frame dropped; sample 1's columns


def test_checkpoint_file_into_full_1024_model_runs_a_step(tmp_path):
    """a Lightning-format .ckpt with the released key prefixes (`model.diffusion_model.*`,
    `first_stage_model.*`, `image_proj_model.*`, and `cond_stage_model.*` / `embedder.*` when the towers carry weights)
    written to disk in fp16, loaded by load_model_checkpoint (strict) into the model the released 1024 YAML builds,
    then one captured-graph denoising step at a small latent. Values are checked on a sample of tensors against the
    file's content after the bf16 repack."""
    from dynamicrafter_amd.lvdm.models.samplers.ddim import DDIMSampler, FusedRun
    from dynamicrafter_amd.scripts.evaluation.inference import load_model_checkpoint
    from dynamicrafter_amd.utils.utils import instantiate_from_config
    from oracle.weights import tensor_for
    root = os.path.join(os.path.dirname(G), "..", "dynamicrafter_amd", "configs")
    cfg = yaml.safe_load(open(os.path.join(root, "inference_1024_v1.0.yaml")))
    model = instantiate_from_config(cfg["model"])
    keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    prefixes = {k.split(".")[0] for k in keys if "." in k}
    assert {"model", "first_stage_model", "image_proj_model"} <= prefixes
    sd = {}
    for k, s in keys.items():
        v = model.state_dict()[k]
        sd[k] = tensor_for(k, s, seed=21).half() if v.is_floating_point() and v.dim() > 0 and k.split(".")[0] in (
            "model", "first_stage_model", "image_proj_model", "cond_stage_model", "embedder") else v.clone()
    f = tmp_path / "model.ckpt"
    torch.save({"state_dict": sd, "global_step": 1}, f)
    del model
    model = load_model_checkpoint(instantiate_from_config(cfg["model"]), str(f)).to(DEV).eval()
    probe = "model.diffusion_model.output_blocks.5.0.temopral_conv.conv3.3.weight"
    assert torch.equal(model.state_dict()[probe].cpu(), sd[probe].float())
    T_, h, w = 16, 8, 8
    gen = torch.Generator().manual_seed(3)
    cond = {"c_crossattn": [torch.randn(1, 77 + 16 * T_, 1024, generator=gen).to(DEV)],
            "c_concat": [(torch.randn(1, 4, T_, h, w, generator=gen) * 0.2).to(DEV)]}
    uc = {"c_crossattn": [torch.randn(1, 77 + 16 * T_, 1024, generator=gen).to(DEV)], "c_concat": cond["c_concat"]}
    s = DDIMSampler(model)
    s.make_schedule(50, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    img = torch.randn(1, 4, T_, h, w, generator=gen).to(DEV)
    run = FusedRun(s, img, [cond, uc], fs=torch.tensor([10], device=DEV), noises=torch.randn(50, 1, 4, T_, h, w, generator=gen).to(DEV),
                   cfg_scale=7.5, guidance_rescale=0.7).capture()
    run.step(); run.step()
    run.sync()
    assert torch.isfinite(run.img).all() and float(run.img.abs().max()) > 0
    # the image projector loaded from the same file runs too
    y = model.image_proj_model(torch.randn(1, 257, 1280, generator=gen).to(DEV))
    assert tuple(y.shape) == (1, 256, 1024) and torch.isfinite(y).all()


def test_bench_two_ranks_share_one_gpu_gloo():
    """`bench.py --gpus 2` itself, launched as the driver launches it (torch.distributed.run, one process per rank): both
    ranks on this box's single GPU, collectives over gloo (DC_BENCH_BACKEND / DC_BENCH_SHARE_GPU) - the code path is the
    one the 8-GPU RCCL run takes, down to scatter_conditioning and the max-over-ranks timing. 256 config, 2 steps."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(G), "..")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DC_BENCH_BACKEND="gloo", DC_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--res", "256", "--no-ae", "--no-trace",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1                                        # rank 0 prints ONE JSON line
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["outputs_finite"]
    assert len(out["config"]["per_rank"]) == 2 and "scatter_conditioning" in out["config"]["conditioning"]
    assert abs(out["value"] - 2 * 16 / out["config"]["clip_seconds"]) < 1e-3 * out["value"]
    assert out["n_ranks_seen"] == 2 and out["backend"] == "gloo"


def test_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2` run DIRECTLY (no launcher, WORLD_SIZE unset), the way the driver runs `--gpus 1`: bench.py
    must start torch.distributed.run itself as a child process, relay rank 0's one JSON line and the return code. Rehearsed on
    this box's single GPU with gloo collectives (DC_BENCH_BACKEND / DC_BENCH_SHARE_GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(G), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DC_BENCH_BACKEND="gloo", DC_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--res", "256", "--no-ae", "--no-trace", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["backend"] == "gloo"
    assert out["config"]["outputs_finite"] and len(out["config"]["per_rank"]) == 2


def test_bench_one_rank_over_rccl():
    """The collectives of the N > 1 job on RCCL itself: `DC_BENCH_FORCE_DIST=1` makes a one-rank bench.py take the N > 1 code path
    with backend "nccl" (= RCCL) - init_process_group(device_id=), broadcast_object_list + dist.scatter of device tensors
    (parallel.scatter_conditioning), barriers, all_reduce(MAX) and all_gather of the timings on the GPU. A one-GPU box cannot host
    two RCCL ranks; with one rank every call still goes through RCCL's API and stream semantics (an argument RCCL rejects, or a
    collective issued on the wrong device, fails here and not first on the 8-GPU node). 256 config, 2 steps."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(G), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "DC_BENCH_BACKEND", "DC_BENCH_SHARE_GPU")}
    env.update(DC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--res", "256", "--no-ae", "--no-trace", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    out = json.loads(line[0])
    assert out["n_gpus"] == 1 and out["n_ranks_seen"] == 1 and out["backend"] == "nccl"
    assert "scatter_conditioning from rank 0 over nccl" in out["config"]["conditioning"]
    assert out["config"]["outputs_finite"] and len(out["config"]["per_rank"]) == 1


def test_ae_frames_per_call_keeps_per_frame_results():
    """`perframe_ae`: the reference calls the AutoencoderKL once per frame (ddpm3d.py:633-639,657-663); here several
    frames ride one launch sequence (GroupNorm / mid attention are per frame, so only the rows per kernel change).
    Same per-frame posterior noise draws, same results up to the bf16 rounding of differently tiled GEMMs."""
    model = _tiny_lvd("inference_512_v1.0.yaml")
    assert model.perframe_ae
    g = torch.Generator().manual_seed(31)
    vid = (torch.rand(1, 3, 6, 64, 96, generator=g) * 2 - 1).to(DEV)
    outs = []
    for k in (1, 4):
        model.ae_frames_per_call = k
        torch.manual_seed(77)                               # the CPU generator the posterior noise is drawn from
        z = model.encode_first_stage(vid)
        rec = model.decode_first_stage(z)
        outs.append((z.clone(), rec.clone(), torch.rand(1).item()))
    assert outs[0][2] == outs[1][2]                         # the same number of draws was consumed
    rz, rr = rel_l2(outs[1][0], outs[0][0]), rel_l2(outs[1][1], outs[0][1])
    print(f"\n[ae frames/call 4 vs 1] latent rel-L2 {rz:.2e}, decoded rel-L2 {rr:.2e}")
    assert rz < 2e-2 and rr < 2e-2


TINY_CLIP = dict(embed_dim=64, vision=dict(image_size=56, layers=3, width=320, heads=4, patch_size=14, mlp_ratio=4.0),
                 text=dict(context_length=77, vocab_size=300, width=128, heads=2, layers=4, mlp_ratio=4.0))


@pytest.mark.parametrize("arch", ["tiny", "ViT-H-14"])
def test_openclip_towers_vs_oracle(arch):
    """the OpenCLIP text tower (causal, penultimate layer, ln_final) and vision tower (bicubic-224
    preprocessing, patch GEMM, 16 heads x 80) on the HIP path against oracle/clip.py with recipe weights - a narrow
    configuration and the released ViT-H/14 widths/depths. (The oracle itself is parity-unpinned against open_clip,
    which is not in the image; see oracle/clip.py.)"""
    from dynamicrafter_amd.lvdm.modules.encoders.condition import (ARCHS, FrozenOpenCLIPEmbedder,
                                                                    FrozenOpenCLIPImageEmbedderV2)
    from oracle import clip as oclip
    from oracle.weights import fill_state_dict
    a = TINY_CLIP if arch == "tiny" else ARCHS[arch]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    txt = FrozenOpenCLIPEmbedder(arch=a, layer="penultimate")
    sd = fill_state_dict({k: tuple(v.shape) for k, v in txt.state_dict().items()}, seed=15)
    sd["model.positional_embedding"] = sd["model.positional_embedding"] * 0.02 / sd["model.positional_embedding"].std()
    txt.load_state_dict(sd, strict=True)
    txt.to(DEV)
    g = torch.Generator().manual_seed(8)
    tokens = torch.randint(0, a["text"]["vocab_size"], (2, 77), generator=g)
    y = txt(tokens.to(DEV))
    ref = oclip.text_forward(sd, tokens, heads=a["text"]["heads"], layer_idx=1)
    r, c = rel_l2(y, ref), cosine(y, ref)
    print(f"\n[openclip {arch}] text tower rel-L2 {r:.3e} cosine {c:.6f}")
    assert tuple(y.shape) == (2, 77, a["text"]["width"]) and r < 2.5e-2 and c > 0.9997    # measured 1.2e-2 (ViT-H/14)
    del txt
    vis = FrozenOpenCLIPImageEmbedderV2(arch=a)
    sd = fill_state_dict({k: tuple(v.shape) for k, v in vis.state_dict().items()}, seed=16)
    vis.load_state_dict(sd, strict=True)
    vis.to(DEV)
    img = torch.rand(2, 3, 320, 512, generator=g) * 2 - 1
    y = vis(img.to(DEV))
    sz = (a["vision"]["image_size"],) * 2
    ref = oclip.vision_forward(sd, img, heads=a["vision"]["heads"], size=sz)
    r, c = rel_l2(y, ref), cosine(y, ref)
    print(f"[openclip {arch}] vision tower rel-L2 {r:.3e} cosine {c:.6f}")
    n_tok = (a["vision"]["image_size"] // a["vision"]["patch_size"]) ** 2 + 1
    assert tuple(y.shape) == (2, n_tok, a["vision"]["width"]) and r < 2.5e-2 and c > 0.9997   # measured 1.2e-2
