"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference) on CPU.

Run in the build container only:  python tests/golden/make_golden.py [--only name,...]
The reference cannot travel to the GPU box, so only its inputs/outputs are committed (as data); weights are
regenerated on both sides from oracle/weights.py's per-name seeded recipe.

Stubs needed to import the reference offline: cv2 (imported by utils/utils.py:3, never called),
pytorch_lightning (LightningModule -> nn.Module + .device), torchvision.utils.make_grid (never called),
OmegaConf -> a plain attribute dict.  DDIMSampler.register_buffer hard-codes .to("cuda") (ddim.py:18-22): the
CPU run overrides it.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def install_stubs():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    pl.LightningModule = LightningModule
    plu = types.ModuleType("pytorch_lightning.utilities")
    plu.rank_zero_only = lambda f: f
    pl.utilities = plu
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.utilities"] = plu
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    tvt = types.ModuleType("torchvision.transforms")      # imported by scripts/evaluation/inference.py, unused on the path
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tvu
    sys.modules["torchvision.transforms"] = tvt
    pl.seed_everything = lambda seed: torch.manual_seed(seed)
    oc = types.ModuleType("omegaconf")                      # idem (only run_inference reads the YAML through it)
    oc.OmegaConf = type("OmegaConf", (), {})
    sys.modules.setdefault("omegaconf", oc)
    if "PIL" not in sys.modules:
        try:
            import PIL  # noqa: F401
        except ImportError:
            pil = types.ModuleType("PIL"); pil.Image = types.ModuleType("PIL.Image")
            sys.modules["PIL"] = pil; sys.modules["PIL.Image"] = pil.Image
    sys.path.insert(0, REF)


class AttrDict(dict):
    """Stand-in for an OmegaConf node: attribute access + dict protocol."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def to_attr(o):
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, list):
        return [to_attr(v) for v in o]
    return o


from oracle.weights import fill_state_dict  # noqa: E402


def load_recipe_weights(module, seed, prefix=""):
    sd = module.state_dict()
    new = fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict(new, strict=True)
    return {k: tuple(v.shape) for k, v in sd.items()}


def rnd(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


TINY_UNET = dict(in_channels=8, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                 channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1, context_dim=128,
                 use_linear=True, use_checkpoint=False, temporal_conv=True, temporal_attention=True,
                 temporal_selfatt_only=True, use_relative_position=False, use_causal_attention=False,
                 temporal_length=4, addition_attention=True, image_cross_attention=True, default_fs=10,
                 fs_condition=True)
TINY_AE = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
               num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------------
def gen_unet_tiny():
    from lvdm.modules.networks.openaimodel3d import UNetModel
    for tag, extra in (("v1024", dict()), ("v256", dict(image_cross_attention_scale_learnable=True, default_fs=3))):
        params = dict(TINY_UNET, **extra)
        net = UNetModel(**params).eval()
        shapes = load_recipe_weights(net, seed=11)
        b, t, h, w = 2, 4, 16, 16
        x = rnd(b, 8, t, h, w, seed=21)
        ctx = rnd(b, 77 + t * 16, params["context_dim"], seed=22)
        ts = torch.tensor([981, 40], dtype=torch.long)
        fs = torch.tensor([24, 3], dtype=torch.long)
        with torch.no_grad():
            y = net(x, ts, context=ctx, fs=fs)
            y_nofs = net(x, ts, context=ctx)          # default_fs path
        save(f"unet_tiny_{tag}", x=x, context=ctx, timesteps=ts.numpy(), fs=fs.numpy(), y=y, y_default_fs=y_nofs,
             param_names=np.array(sorted(shapes)), yaml_params=np.array(yaml.safe_dump(params)))


def gen_unet_fullwidth():
    """Real channel widths of the released configs at an 8x8 latent, T=16 (weights from the recipe, not stored)."""
    from lvdm.modules.networks.openaimodel3d import UNetModel
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/inference_1024_v1.0.yaml")))
    params = cfg["model"]["params"]["unet_config"]["params"]
    params["use_checkpoint"] = False
    net = UNetModel(**params).eval()
    shapes = load_recipe_weights(net, seed=12)
    b, t, h, w = 1, 16, 8, 8
    x = rnd(b, 8, t, h, w, seed=31)
    ctx = rnd(b, 77 + t * 16, 1024, seed=32)
    ts = torch.tensor([500], dtype=torch.long)
    fs = torch.tensor([10], dtype=torch.long)
    with torch.no_grad():
        y = net(x, ts, context=ctx, fs=fs)
    n_params = sum(int(np.prod(s)) for s in shapes.values())
    save("unet_fullwidth_8x8", x=x, context=ctx, timesteps=ts.numpy(), fs=fs.numpy(), y=y,
         n_params=np.array(n_params), n_tensors=np.array(len(shapes)),
         key_digest=np.array(sorted(f"{k}:{'x'.join(map(str, s))}" for k, s in shapes.items())))


def gen_ae():
    from lvdm.models.autoencoder import AutoencoderKL
    for tag, dd in (("tiny", TINY_AE), ("full", dict(TINY_AE, ch=128))):
        ae = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4).eval()
        shapes = load_recipe_weights(ae, seed=13)
        img = rnd(2, 3, 64, 96, seed=41).clamp(-1, 1)
        noise = rnd(2, 4, 8, 12, seed=42)
        with torch.no_grad():
            post = ae.encode(img)
            z = post.sample(noise=noise)
            zmode = post.mode()
            rec = ae.decode(z)
        save(f"ae_{tag}", img=img, noise=noise, moments=post.parameters, z=z, z_mode=zmode, rec=rec,
             param_names=np.array(sorted(shapes)), yaml_params=np.array(yaml.safe_dump(dd)))


def build_lvd(config_name, unet_params=None, ae_dd=None, conditioners=None):
    """LatentVisualDiffusion from a released YAML with small nets and Identity conditioners."""
    from utils.utils import instantiate_from_config
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", config_name)))
    m = cfg["model"]
    p = m["params"]
    if unet_params is not None:
        keep = {k: p["unet_config"]["params"][k] for k in ("default_fs",) if k in p["unet_config"]["params"]}
        p["unet_config"]["params"] = dict(unet_params, **keep)
        if p["unet_config"]["params"].get("image_cross_attention_scale_learnable") is None:
            pass
    p["unet_config"]["params"]["use_checkpoint"] = False
    if ae_dd is not None:
        p["first_stage_config"]["params"]["ddconfig"] = dict(ae_dd)
    p["cond_stage_config"] = {"target": "torch.nn.Identity"}
    p["img_cond_stage_config"] = {"target": "torch.nn.Identity"}
    p["image_proj_stage_config"] = {"target": "torch.nn.Identity"}
    p.update(conditioners or {})
    model = instantiate_from_config(to_attr(m)).eval()
    return model, p


def cpu_sampler_cls(base):
    class CpuSampler(base):
        def register_buffer(self, name, attr):     # reference hard-codes .to("cuda")
            setattr(self, name, attr)
    return CpuSampler


def gen_schedules():
    from lvdm.models.samplers.ddim import DDIMSampler
    Sampler = cpu_sampler_cls(DDIMSampler)
    out = {}
    for cname in ("inference_256_v1.0.yaml", "inference_512_v1.0.yaml", "inference_1024_v1.0.yaml"):
        model, p = build_lvd(cname, TINY_UNET, TINY_AE)
        tag = cname.split("_")[1]
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod"):
            out[f"{tag}/{k}"] = getattr(model, k).numpy()
        if model.use_dynamic_rescale:
            out[f"{tag}/scale_arr"] = model.scale_arr.numpy()
        for S in (10, 50):
            for disc in ("uniform", "uniform_trailing"):
                for eta in (0.0, 1.0):
                    s = Sampler(model)
                    s.make_schedule(S, ddim_discretize=disc, ddim_eta=eta, verbose=False)
                    key = f"{tag}/S{S}/{disc}/eta{int(eta)}"
                    out[key + "/ddim_timesteps"] = np.asarray(s.ddim_timesteps)
                    # the values the arithmetic sees: torch.full(size, v) casts to fp32 (ddim.py:251-254)
                    for nm, arr in (("a_t", s.ddim_alphas), ("a_prev", s.ddim_alphas_prev), ("sigma_t", s.ddim_sigmas),
                                    ("sqrt_one_minus_at", s.ddim_sqrt_one_minus_alphas)):
                        out[f"{key}/{nm}"] = np.array([torch.full((1,), arr[i]).item() for i in range(S)], dtype=np.float32)
                    if model.use_dynamic_rescale:
                        out[f"{key}/scale_t"] = np.array([torch.full((1,), s.ddim_scale_arr[i]).item() for i in range(S)], dtype=np.float32)
                        out[f"{key}/scale_prev"] = np.array([torch.full((1,), s.ddim_scale_arr_prev[i]).item() for i in range(S)], dtype=np.float32)
                    out[f"{key}/sigma_raw_f64"] = np.asarray(s.ddim_sigmas, dtype=np.float64)
    save("schedules", **out)


class FakeModel:
    """Carries a real model's schedule buffers; apply_model returns queued tensors."""
    def __init__(self, real, outputs):
        self.__dict__.update({k: getattr(real, k) for k in (
            "num_timesteps", "betas", "alphas_cumprod", "alphas_cumprod_prev", "parameterization",
            "use_dynamic_rescale", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")})
        if real.use_dynamic_rescale:
            self.scale_arr = real.scale_arr
        self.device = torch.device("cpu")
        self.predict_eps_from_z_and_v = real.predict_eps_from_z_and_v
        self.predict_start_from_z_and_v = real.predict_start_from_z_and_v
        self._outs = list(outputs)

    def apply_model(self, x, t, c, **kw):
        return self._outs.pop(0)


def gen_p_sample_known_answers():
    import lvdm.models.samplers.ddim as ddim_mod
    import lvdm.models.samplers.ddim_multiplecond as mc_mod
    Sampler = cpu_sampler_cls(ddim_mod.DDIMSampler)
    SamplerMC = cpu_sampler_cls(mc_mod.DDIMSampler)
    out = {}
    shape = (2, 4, 4, 6, 5)
    for cname, disc, gr in (("inference_256_v1.0.yaml", "uniform", 0.0), ("inference_512_v1.0.yaml", "uniform_trailing", 0.7),
                            ("inference_1024_v1.0.yaml", "uniform_trailing", 0.7)):
        model, p = build_lvd(cname, TINY_UNET, TINY_AE)
        tag = cname.split("_")[1]
        for eta in (0.0, 1.0):
            for index in (9, 4, 0):
                x = rnd(*shape, seed=51); ec = rnd(*shape, seed=52); eu = rnd(*shape, seed=53) * 0.9 + 0.1 * ec
                ei = rnd(*shape, seed=55) * 0.5 + 0.5 * ec
                noise = rnd(*shape, seed=54)
                for mod, S_cls, extra, nm in ((ddim_mod, Sampler, {}, "cfg2"),
                                              (mc_mod, SamplerMC, dict(cfg_img=2.0, unconditional_conditioning_img_nonetext={"c": 2}), "cfg3")):
                    outs = [ec, eu] + ([ei] if nm == "cfg3" else [])
                    fm = FakeModel(model, outs)
                    s = S_cls(fm)
                    s.make_schedule(10, ddim_discretize=disc, ddim_eta=eta, verbose=False)
                    mod.noise_like = lambda shp, dev, rep=False: noise      # injected noise
                    step = int(np.flip(s.ddim_timesteps)[10 - index - 1])
                    ts = torch.full((shape[0],), step, dtype=torch.long)
                    xp, px0 = s.p_sample_ddim(x.clone(), {"c": 1}, ts, index=index, unconditional_guidance_scale=7.5,
                                              unconditional_conditioning={"c": 0}, guidance_rescale=gr, **extra)
                    key = f"{tag}/{disc}/eta{int(eta)}/i{index}/{nm}"
                    out[key + "/x_prev"] = xp.numpy(); out[key + "/pred_x0"] = px0.numpy()
    out["x"] = x.numpy(); out["e_cond"] = ec.numpy(); out["e_uncond"] = eu.numpy(); out["e_img"] = ei.numpy()
    out["noise"] = noise.numpy()
    save("p_sample_ddim", **out)


def gen_trajectory():
    """Whole sampler + apply_model('hybrid') + tiny UNet: 10 steps with CFG, eps-param (256 cfg) and v-param +
    ZTSNR + dynamic rescale + guidance rescale + eta=1 with injected noise (512 cfg)."""
    import lvdm.models.samplers.ddim as ddim_mod
    Sampler = cpu_sampler_cls(ddim_mod.DDIMSampler)
    for cname, disc, eta, gr, extra in (
            ("inference_256_v1.0.yaml", "uniform", 0.0, 0.0, dict(image_cross_attention_scale_learnable=True)),
            ("inference_512_v1.0.yaml", "uniform_trailing", 1.0, 0.7, dict())):
        tag = cname.split("_")[1]
        model, p = build_lvd(cname, dict(TINY_UNET, **extra), TINY_AE)
        load_recipe_weights(model.model.diffusion_model, seed=11)
        b, t, h, w = 1, 4, 16, 16
        S = 10
        x_T = rnd(b, 4, t, h, w, seed=61)
        cond = {"c_crossattn": [rnd(b, 77 + 16 * t, 128, seed=62)], "c_concat": [rnd(b, 4, t, h, w, seed=63) * 0.18215]}
        uc = {"c_crossattn": [rnd(b, 77 + 16 * t, 128, seed=64)], "c_concat": cond["c_concat"]}
        noises = [rnd(b, 4, t, h, w, seed=70 + i) for i in range(S)]
        it = iter(noises)
        ddim_mod.noise_like = lambda shp, dev, rep=False: next(it)
        s = Sampler(model)
        fs = torch.tensor([p["unet_config"]["params"]["default_fs"]] * b, dtype=torch.long)
        samples, inter = s.sample(S=S, batch_size=b, shape=(4, t, h, w), conditioning=cond, verbose=False,
                                  unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=eta,
                                  x_T=x_T, fs=fs, timestep_spacing=disc, guidance_rescale=gr, log_every_t=1)
        save(f"trajectory_{tag}", x_T=x_T, ctx=cond["c_crossattn"][0], uc_ctx=uc["c_crossattn"][0],
             c_concat=cond["c_concat"][0], noises=torch.stack(noises), fs=fs.numpy(), samples=samples,
             x_inter=torch.stack(inter["x_inter"]), yaml_unet=np.array(yaml.safe_dump(p["unet_config"]["params"])))


def gen_first_stage():
    """LatentDiffusion.encode_first_stage / decode_first_stage with perframe_ae, tiny AE."""
    model, p = build_lvd("inference_512_v1.0.yaml", TINY_UNET, TINY_AE)
    load_recipe_weights(model.first_stage_model, seed=13)
    import lvdm.distributions as dist
    vid = rnd(1, 3, 3, 32, 48, seed=81).clamp(-1, 1)
    noise = [rnd(1, 4, 4, 6, seed=90 + i) for i in range(3)]
    it = iter(noise)
    orig = torch.randn
    torch.randn = lambda *a, **k: next(it)          # DiagonalGaussianDistribution.sample draws torch.randn(shape)
    try:
        z = model.encode_first_stage(vid)
    finally:
        torch.randn = orig
    rec = model.decode_first_stage(z)
    save("first_stage", video=vid, noise=torch.cat(noise, 0), z=z, rec=rec, scale_factor=np.array(model.scale_factor))


def gen_harness():
    """scripts/evaluation/inference.py:image_guided_synthesis of the reference, driven end to end on CPU: toy CLIP
    stand-ins (tests/golden_cfg.py) -> reference Resampler -> conditioning assembly -> DDIM loop -> per-frame AE
    decode. Cases: (a) 512 config, 2-branch CFG 7.5, guidance rescale, eta=1, text prompt; (b) 256 config, interp
    mode (first/last-frame concat), 3-branch guidance (multiple_cond_cfg, cfg_img=2), eta=0."""
    import importlib.util
    from tests.golden_cfg import TINY_RESAMPLER
    spec = importlib.util.spec_from_file_location("ref_inference", os.path.join(REF, "scripts", "evaluation", "inference.py"))
    inf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(inf)
    import lvdm.models.samplers.ddim as ddim_mod
    import lvdm.models.samplers.ddim_multiplecond as mc_mod
    inf.DDIMSampler = cpu_sampler_cls(ddim_mod.DDIMSampler)
    inf.DDIMSampler_multicond = cpu_sampler_cls(mc_mod.DDIMSampler)
    for tag, cname, kw in (
            ("a", "inference_512_v1.0.yaml", dict(ddim_steps=5, ddim_eta=1.0, unconditional_guidance_scale=7.5, fs=24,
                                                  text_input=True, timestep_spacing="uniform_trailing", guidance_rescale=0.7)),
            ("b", "inference_256_v1.0.yaml", dict(ddim_steps=4, ddim_eta=0.0, unconditional_guidance_scale=7.5, fs=3,
                                                  text_input=False, multiple_cond_cfg=True, cfg_img=2.0, interp=True,
                                                  timestep_spacing="uniform", guidance_rescale=0.0))):
        extra = dict(image_cross_attention_scale_learnable=True) if "256" in cname else {}
        model, p = build_lvd(cname, dict(TINY_UNET, **extra), TINY_AE,
                             conditioners=dict(cond_stage_config={"target": "tests.golden_cfg.ToyTextEmbedder"},
                                               img_cond_stage_config={"target": "tests.golden_cfg.ToyImageEmbedder"},
                                               image_proj_stage_config={"target": "lvdm.modules.encoders.resampler.Resampler",
                                                                        "params": dict(TINY_RESAMPLER)}))
        load_recipe_weights(model.model.diffusion_model, seed=11)
        load_recipe_weights(model.first_stage_model, seed=13)
        load_recipe_weights(model.image_proj_model, seed=14)
        b, t, H, W = 1, 4, 64, 64
        h, w = H // 8, W // 8
        videos = rnd(b, 3, t, H, W, seed=101).clamp(-1, 1)
        if kw.get("interp"):
            videos[:, :, 1:-1] = 0.0
        x_T = rnd(b, 4, t, h, w, seed=102)
        S = kw["ddim_steps"]
        noises = [rnd(b, 4, t, h, w, seed=110 + i) for i in range(S)]
        ae_noise = [rnd(1, 4, h, w, seed=120 + i) for i in range(b * t)]
        it_n, it_a = iter(noises), iter(ae_noise)
        ddim_mod.noise_like = lambda shp, dev, rep=False: next(it_n)
        mc_mod.noise_like = ddim_mod.noise_like
        orig = torch.randn
        torch.randn = lambda *a, **k: next(it_a)        # only DiagonalGaussianDistribution.sample draws it (x_T is injected)
        try:
            with torch.no_grad():
                out = inf.image_guided_synthesis(model, ["a corgi running on the beach"], videos, [b, 4, t, h, w],
                                                 n_samples=1, x_T=x_T, **kw)
        finally:
            torch.randn = orig
        assert out.shape == (b, 1, 3, t, H, W), out.shape
        save(f"harness_{tag}", videos=videos, x_T=x_T, noises=torch.stack(noises), ae_noise=torch.cat(ae_noise, 0), out=out,
             kwargs=np.array(yaml.safe_dump(kw)))


def gen_resampler():
    """Reference Resampler: a narrow one and the released configuration (dim 1024, depth 4, 12 heads, 16 queries x 16
    frames) with recipe weights."""
    from lvdm.modules.encoders.resampler import Resampler
    for tag, kw, n1 in (("tiny", dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=4, embedding_dim=64,
                                      output_dim=128, ff_mult=4, video_length=4), 9),
                        ("full", dict(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280,
                                      output_dim=1024, ff_mult=4, video_length=16), 257)):
        m = Resampler(**kw).eval()
        shapes = load_recipe_weights(m, seed=14)
        x = rnd(2 if tag == "tiny" else 1, n1, kw["embedding_dim"], seed=95)
        with torch.no_grad():
            y = m(x)
        save(f"resampler_{tag}", x=x, y=y, param_names=np.array(sorted(shapes)), yaml_params=np.array(yaml.safe_dump(kw)))



def gen_unet_fullsize():
    """The released 1.44 B-parameter UNet run by the REFERENCE at the real latent sizes of BASELINE.json configs 1 and
    2/5: inference_256 (eps, learnable alpha, fs 3) at 16x32x32 and inference_512 at 16x40x64 with the interp (config 5)
    conditioning pattern - the concat half of the input is zero except frames 0 and 15 (inference.py:246-249), fs 5.
    (72x128 cannot run through the reference's plain attention on 64 GB: 80 x 9216^2 fp32 scores; that size is
    covered by the chunked oracle, which these two fixtures pin at full width and real token counts.)"""
    from lvdm.modules.networks.openaimodel3d import UNetModel
    for tag, cname, (h, w), fsv, interp in (("256_32x32", "inference_256_v1.0.yaml", (32, 32), 3, False),
                                            ("512_40x64_interp", "inference_512_v1.0.yaml", (40, 64), 5, True)):
        cfg = yaml.safe_load(open(os.path.join(REF, "configs", cname)))
        params = cfg["model"]["params"]["unet_config"]["params"]
        params["use_checkpoint"] = False
        net = UNetModel(**params).eval()
        load_recipe_weights(net, seed=12)
        b, t = 1, 16
        xn = rnd(b, 4, t, h, w, seed=131)
        cc = rnd(b, 4, t, h, w, seed=132) * 0.18215 * 4
        if interp:
            cc[:, :, 1:-1] = 0.0
        else:
            cc = cc[:, :, :1].repeat(1, 1, t, 1, 1)
        ctx = rnd(b, 77 + t * 16, 1024, seed=133)
        ts = torch.tensor([759], dtype=torch.long)
        fs = torch.tensor([fsv], dtype=torch.long)
        import time
        t0 = time.time()
        with torch.no_grad():
            y = net(torch.cat([xn, cc], 1), ts, context=ctx, fs=fs)
        print(f"  reference forward {tag}: {time.time() - t0:.1f} s")
        # inputs are stored in fp16-exact form? no: store seeds' tensors as they are (fp32) - a few MB
        save(f"unet_fullsize_{tag}", x=xn, c_concat=cc, context=ctx, timesteps=ts.numpy(), fs=fs.numpy(), y=y,
             yaml_params=np.array(yaml.safe_dump(params)))
        del net


def gen_trajectory50():
    """The bench runs 50 eta=1 steps; the 10-step fixtures cannot show drift over that length. Same tiny network and
    settings as trajectory_512 (v-param + ZTSNR + dynamic rescale + guidance rescale 0.7, uniform_trailing), S=50."""
    import lvdm.models.samplers.ddim as ddim_mod
    Sampler = cpu_sampler_cls(ddim_mod.DDIMSampler)
    model, p = build_lvd("inference_512_v1.0.yaml", dict(TINY_UNET), TINY_AE)
    load_recipe_weights(model.model.diffusion_model, seed=11)
    b, t, h, w = 1, 4, 16, 16
    S = 50
    x_T = rnd(b, 4, t, h, w, seed=161)
    cond = {"c_crossattn": [rnd(b, 77 + 16 * t, 128, seed=162)], "c_concat": [rnd(b, 4, t, h, w, seed=163) * 0.18215]}
    uc = {"c_crossattn": [rnd(b, 77 + 16 * t, 128, seed=164)], "c_concat": cond["c_concat"]}
    noises = [rnd(b, 4, t, h, w, seed=1700 + i) for i in range(S)]
    it = iter(noises)
    ddim_mod.noise_like = lambda shp, dev, rep=False: next(it)
    s = Sampler(model)
    fs = torch.tensor([p["unet_config"]["params"]["default_fs"]] * b, dtype=torch.long)
    samples, inter = s.sample(S=S, batch_size=b, shape=(4, t, h, w), conditioning=cond, verbose=False,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0,
                              x_T=x_T, fs=fs, timestep_spacing="uniform_trailing", guidance_rescale=0.7, log_every_t=10)
    save("trajectory50_512", x_T=x_T, ctx=cond["c_crossattn"][0], uc_ctx=uc["c_crossattn"][0],
         c_concat=cond["c_concat"][0], noises=torch.stack(noises), fs=fs.numpy(), samples=samples,
         x_inter=torch.stack(inter["x_inter"]), noise_seeds=np.array([1700 + i for i in range(S)]))


def gen_sampler_extras():
    """SURVEY 8(f) rank 2 remainder, from the reference sampler on the tiny 512-config model: (1) mask / x0 blending
    inside ddim_sampling (ddim.py:174-180; q_sample noise injected through torch.randn_like), also with clean_cond;
    (2) decode() from an intermediate latent (:281-301); (3) stochastic_encode() (:303-317)."""
    import lvdm.models.samplers.ddim as ddim_mod
    Sampler = cpu_sampler_cls(ddim_mod.DDIMSampler)
    model, p = build_lvd("inference_512_v1.0.yaml", dict(TINY_UNET), TINY_AE)
    load_recipe_weights(model.model.diffusion_model, seed=11)
    b, t, h, w = 1, 4, 16, 16
    S = 6
    x_T = rnd(b, 4, t, h, w, seed=181)
    x0 = rnd(b, 4, t, h, w, seed=182)
    mask = (rnd(b, 1, t, h, w, seed=183) > 0).float().expand(b, 4, t, h, w).contiguous()
    cond = {"c_crossattn": [rnd(b, 77 + 16 * t, 128, seed=184)], "c_concat": [rnd(b, 4, t, h, w, seed=185) * 0.18215]}
    uc = {"c_crossattn": [rnd(b, 77 + 16 * t, 128, seed=186)], "c_concat": cond["c_concat"]}
    noises = [rnd(b, 4, t, h, w, seed=190 + i) for i in range(S)]
    qnoises = [rnd(b, 4, t, h, w, seed=200 + i) for i in range(S)]
    fs = torch.tensor([24] * b, dtype=torch.long)
    out = dict(x_T=x_T, x0=x0, mask=mask, ctx=cond["c_crossattn"][0], uc_ctx=uc["c_crossattn"][0],
               c_concat=cond["c_concat"][0], noises=torch.stack(noises), qnoises=torch.stack(qnoises), fs=fs.numpy())
    orig_rl = torch.randn_like
    for tag, clean in (("mask", False), ("mask_clean", True)):
        it_n, it_q = iter(noises), iter(qnoises)
        ddim_mod.noise_like = lambda shp, dev, rep=False: next(it_n)
        torch.randn_like = lambda x, **k: next(it_q)
        try:
            s = Sampler(model)
            samples, _ = s.sample(S=S, batch_size=b, shape=(4, t, h, w), conditioning=cond, verbose=False, mask=mask,
                                  x0=x0, unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=1.0,
                                  x_T=x_T, fs=fs, timestep_spacing="uniform_trailing", guidance_rescale=0.7,
                                  clean_cond=clean)
        finally:
            torch.randn_like = orig_rl
        out[f"{tag}/samples"] = samples
    # decode(): last t_start DDIM steps from a latent, eta=0 schedule, default fs (decode passes no fs)
    s = Sampler(model)
    s.make_schedule(S, ddim_discretize="uniform", ddim_eta=0.0, verbose=False)
    ddim_mod.noise_like = lambda shp, dev, rep=False: torch.zeros(shp)
    x_lat = rnd(b, 4, t, h, w, seed=187)
    out["decode/x_latent"] = x_lat
    out["decode/t_start"] = np.array(4)
    out["decode/x_dec"] = s.decode(x_lat, cond, 4, unconditional_guidance_scale=7.5, unconditional_conditioning=uc)
    # stochastic_encode(): both table choices
    tt = torch.tensor([3], dtype=torch.long)
    n_enc = rnd(b, 4, t, h, w, seed=188)
    out["enc/t"] = tt.numpy(); out["enc/noise"] = n_enc
    out["enc/ddim"] = s.stochastic_encode(x0, tt, use_original_steps=False, noise=n_enc)
    tt2 = torch.tensor([640], dtype=torch.long)
    out["enc/t_orig"] = tt2.numpy()
    out["enc/orig"] = s.stochastic_encode(x0, tt2, use_original_steps=True, noise=n_enc)
    save("sampler_extras", **out)


def gen_harness_interp512():
    """BASELINE.json config 5 ('512_interp'): the 512 YAML (v-param, ZTSNR, dynamic rescale) driven as
    scripts/run_application.sh:8-28 does - interp=True (first + last frame concat), fs=5, uniform_trailing,
    guidance_rescale 0.7, eta=1 - through the reference's image_guided_synthesis; tiny nets, 5 steps."""
    import importlib.util
    from tests.golden_cfg import TINY_RESAMPLER
    spec = importlib.util.spec_from_file_location("ref_inference", os.path.join(REF, "scripts", "evaluation", "inference.py"))
    inf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(inf)
    import lvdm.models.samplers.ddim as ddim_mod
    import lvdm.models.samplers.ddim_multiplecond as mc_mod
    inf.DDIMSampler = cpu_sampler_cls(ddim_mod.DDIMSampler)
    inf.DDIMSampler_multicond = cpu_sampler_cls(mc_mod.DDIMSampler)
    kw = dict(ddim_steps=5, ddim_eta=1.0, unconditional_guidance_scale=7.5, fs=5, text_input=True, interp=True,
              timestep_spacing="uniform_trailing", guidance_rescale=0.7)
    model, p = build_lvd("inference_512_v1.0.yaml", dict(TINY_UNET), TINY_AE,
                         conditioners=dict(cond_stage_config={"target": "tests.golden_cfg.ToyTextEmbedder"},
                                           img_cond_stage_config={"target": "tests.golden_cfg.ToyImageEmbedder"},
                                           image_proj_stage_config={"target": "lvdm.modules.encoders.resampler.Resampler",
                                                                    "params": dict(TINY_RESAMPLER)}))
    load_recipe_weights(model.model.diffusion_model, seed=11)
    load_recipe_weights(model.first_stage_model, seed=13)
    load_recipe_weights(model.image_proj_model, seed=14)
    b, t, H, W = 1, 4, 64, 128
    h, w = H // 8, W // 8
    videos = rnd(b, 3, t, H, W, seed=141).clamp(-1, 1)
    videos[:, :, 1:-1] = 0.0                                   # load_data_prompts(interp): frames between are unused
    x_T = rnd(b, 4, t, h, w, seed=142)
    S = kw["ddim_steps"]
    noises = [rnd(b, 4, t, h, w, seed=150 + i) for i in range(S)]
    ae_noise = [rnd(1, 4, h, w, seed=160 + i) for i in range(b * t)]
    it_n, it_a = iter(noises), iter(ae_noise)
    ddim_mod.noise_like = lambda shp, dev, rep=False: next(it_n)
    orig = torch.randn
    torch.randn = lambda *a, **k: next(it_a)
    try:
        with torch.no_grad():
            out = inf.image_guided_synthesis(model, ["two frames of a blooming flower"], videos, [b, 4, t, h, w],
                                             n_samples=1, x_T=x_T, **kw)
    finally:
        torch.randn = orig
    assert out.shape == (b, 1, 3, t, H, W), out.shape
    save("harness_c", videos=videos, x_T=x_T, noises=torch.stack(noises), ae_noise=torch.cat(ae_noise, 0), out=out,
         kwargs=np.array(yaml.safe_dump(kw)))


def gen_trajectory_fullwidth():
    """BASELINE.json config 1 as SURVEY 8(d) specifies it, run by the REFERENCE: inference_256_v1.0.yaml with the released
    1.44 B-parameter UNet (recipe weights), latent 16x32x32, 10 DDIM steps, `uniform`, CFG 7.5, injected x_T - once with
    eta = 0 and once with eta = 1 + injected per-step noises. The whole sampler runs (DDIMSampler.sample ->
    apply_model('hybrid') -> UNetModel), cond and uncond as two batch-1 forwards per step as the reference issues them.
    Stored: the inputs, x after steps 1 / 5 / 10 and pred_x0 at the same steps. The context is rounded to fp16 BEFORE the
    reference sees it (stored as fp16, exact); the eta = 1 noises are seeded torch.randn draws (seeds + a checksum stored)."""
    import time
    import lvdm.models.samplers.ddim as ddim_mod
    Sampler = cpu_sampler_cls(ddim_mod.DDIMSampler)
    model, p = build_lvd("inference_256_v1.0.yaml", None, TINY_AE)
    load_recipe_weights(model.model.diffusion_model, seed=12)
    b, t, h, w = 1, 16, 32, 32
    S = 10
    x_T = rnd(b, 4, t, h, w, seed=211)
    ctx = rnd(b, 77 + 16 * t, 1024, seed=212).half().float()
    uctx = rnd(b, 77 + 16 * t, 1024, seed=214).half().float()
    cc = (rnd(b, 4, 1, h, w, seed=213) * 0.18215 * 4).repeat(1, 1, t, 1, 1)
    cond = {"c_crossattn": [ctx], "c_concat": [cc]}
    uc = {"c_crossattn": [uctx], "c_concat": [cc]}
    fs = torch.tensor([p["unet_config"]["params"]["default_fs"]] * b, dtype=torch.long)
    seeds = [2200 + i for i in range(S)]
    out = dict(x_T=x_T, ctx=ctx.half(), uc_ctx=uctx.half(), c_concat=cc[:, :, :1], fs=fs.numpy(),
               noise_seeds=np.array(seeds), keep=np.array([1, 5, 10]),
               yaml_unet=np.array(yaml.safe_dump(p["unet_config"]["params"])))
    for tag, eta in (("eta0", 0.0), ("eta1", 1.0)):
        noises = [rnd(b, 4, t, h, w, seed=sd) for sd in seeds]
        it = iter(noises)
        ddim_mod.noise_like = lambda shp, dev, rep=False: next(it)
        s = Sampler(model)
        t0 = time.time()
        with torch.no_grad():
            samples, inter = s.sample(S=S, batch_size=b, shape=(4, t, h, w), conditioning=cond, verbose=False,
                                      unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=eta,
                                      x_T=x_T, fs=fs, timestep_spacing="uniform", guidance_rescale=0.0, log_every_t=1)
        print(f"  reference 10-step run {tag}: {time.time() - t0:.1f} s")
        xi, pi = inter["x_inter"], inter["pred_x0"]          # index 0 = x_T, index k = after step k
        assert len(xi) == S + 1 and torch.equal(xi[S], samples)
        for k in (1, 5, 10):
            out[f"{tag}/x_{k}"] = xi[k]
            out[f"{tag}/pred_x0_{k}"] = pi[k]
        out[f"{tag}/noise_checksum"] = np.array([float(n.double().sum()) for n in noises])
    save("trajectory_fullwidth_256", **out)


GENS = dict(resampler=gen_resampler, unet_tiny=gen_unet_tiny, unet_fullwidth=gen_unet_fullwidth, ae=gen_ae, schedules=gen_schedules,
            p_sample=gen_p_sample_known_answers, trajectory=gen_trajectory, first_stage=gen_first_stage,
            harness=gen_harness, unet_fullsize=gen_unet_fullsize, trajectory50=gen_trajectory50,
            sampler_extras=gen_sampler_extras, harness_interp512=gen_harness_interp512,
            trajectory_fullwidth=gen_trajectory_fullwidth)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    install_stubs()
    torch.set_num_threads(8)
    names = [n for n in args.only.split(",") if n] or list(GENS)
    for n in names:
        print("==", n)
        GENS[n]()
