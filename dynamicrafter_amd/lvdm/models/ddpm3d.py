"""LatentVisualDiffusion — the inference shell around the HIP UNet and AutoencoderKL.

Drop-in for the inference surface of lvdm/models/ddpm3d.py (DDPM :40, LatentDiffusion :464,
LatentVisualDiffusion :1029, DiffusionWrapper :1237): same constructor keywords as `configs/*.yaml model.params`,
same schedule buffers, `apply_model`, `encode_first_stage` / `decode_first_stage`, `get_learned_conditioning`,
`predict_*_from_z_and_v`, `q_sample`; same state_dict prefixes (`model.diffusion_model.*`, `first_stage_model.*`).
Training-only members (losses, optimizers, EMA, logging) are out of scope.

Additions for the MI355X path (not in the reference): `apply_model_rows` evaluates several conditioning branches
(cond / uncond [/ image-only]) of classifier-free guidance as ONE batched UNet forward over channels-last rows and
hands the raw row output to the fused DDIM kernel; DDIMSampler uses it when present.
"""
import logging
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ...utils.utils import instantiate_from_config
from ..distributions import DiagonalGaussianDistribution
from ..modules.networks.openaimodel3d import C_IN_PAD
from .utils_diffusion import make_beta_schedule, rescale_zero_terminal_snr

mainlogger = logging.getLogger("mainlogger")


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, TypeError):
        return getattr(cfg, key, default)


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


class DiffusionWrapper(nn.Module):
    """Owns `diffusion_model` (state_dict prefix) and assembles the UNet inputs (ddpm3d.py:1243-1305)."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        if conditioning_key not in (None, "concat", "crossattn", "hybrid"):
            raise NotImplementedError(f"conditioning_key '{conditioning_key}' is not used by DynamiCrafter inference")

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, **kwargs):
        net = self.diffusion_model
        if self.conditioning_key in ("hybrid", "concat"):
            # cat([x] + c_concat, dim=1) happens inside the packing kernel (no intermediate tensor)
            cc = c_concat[0] if len(c_concat) == 1 else torch.cat(c_concat, dim=1)
            xc = (x, cc)
        else:
            xc = (x, None)
        ctx = None
        if self.conditioning_key in ("hybrid", "crossattn"):
            ctx = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        return net.forward_pair(xc[0], xc[1], t, ctx, **kwargs)


class DDPM(nn.Module):
    """Schedule buffers + parameterisation helpers (ddpm3d.py:40-308, inference subset)."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None,
                 ignore_keys=(), load_only_unet=False, monitor=None, use_ema=True, first_stage_key="image",
                 image_size=256, channels=3, log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, given_betas=None, original_elbo_weight=0., v_posterior=0., l_simple_weight=1.,
                 conditioning_key=None, parameterization="eps", scheduler_config=None,
                 use_positional_encodings=False, learn_logvar=False, logvar_init=0., rescale_betas_zero_snr=False):
        super().__init__()
        assert parameterization in ("eps", "x0", "v")
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised = clip_denoised
        self.log_every_t = log_every_t
        self.first_stage_key = first_stage_key
        self.channels = channels
        self.temporal_length = _get(_get(unet_config, "params"), "temporal_length")
        self.image_size = [image_size, image_size] if isinstance(image_size, int) else list(image_size)
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = False                       # inference: every released config sets use_ema False
        self.rescale_betas_zero_snr = rescale_betas_zero_snr
        self.v_posterior = v_posterior
        if monitor is not None:
            self.monitor = monitor
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self._ckpt_path, self._ignore_keys, self._load_only_unet = ckpt_path, ignore_keys, load_only_unet

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        if self.rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        abar = np.cumprod(1. - betas, axis=0)
        abar_prev = np.append(1., abar[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = partial(torch.tensor, dtype=torch.float32)
        for name, val in (("betas", betas), ("alphas_cumprod", abar), ("alphas_cumprod_prev", abar_prev),
                          ("sqrt_alphas_cumprod", np.sqrt(abar)),
                          ("sqrt_one_minus_alphas_cumprod", np.sqrt(1. - abar)),
                          ("log_one_minus_alphas_cumprod", np.log(np.maximum(1. - abar, 1e-300)))):
            self.register_buffer(name, f32(val))
        if self.parameterization != "v":
            self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1. / abar)))
            self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1. / abar - 1)))
        else:
            self.register_buffer("sqrt_recip_alphas_cumprod", torch.zeros(self.num_timesteps))
            self.register_buffer("sqrt_recipm1_alphas_cumprod", torch.zeros(self.num_timesteps))
        with np.errstate(divide="ignore", invalid="ignore"):
            post_var = (1 - self.v_posterior) * betas * (1. - abar_prev) / (1. - abar) + self.v_posterior * betas
            self.register_buffer("posterior_variance", f32(post_var))
            self.register_buffer("posterior_log_variance_clipped", f32(np.log(np.maximum(post_var, 1e-20))))
            self.register_buffer("posterior_mean_coef1", f32(betas * np.sqrt(abar_prev) / (1. - abar)))
            self.register_buffer("posterior_mean_coef2", f32((1. - abar_prev) * np.sqrt(1. - betas) / (1. - abar)))

    # parameterisation helpers on [B, ...] tensors (torch elementwise; the sampler's hot path uses dc_ddim_step)
    def predict_start_from_z_and_v(self, x_t, t, v):
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_t.shape) * x_t -
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_t.shape) * v)

    def predict_eps_from_z_and_v(self, x_t, t, v):
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_t.shape) * v +
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_t.shape) * x_t)

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t -
                extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def init_from_ckpt(self, path, ignore_keys=(), only_model=False):
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        target = self.model if only_model else self
        missing, unexpected = target.load_state_dict(sd, strict=False)
        mainlogger.info(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")


class LatentDiffusion(DDPM):
    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="caption",
                 cond_stage_trainable=False, cond_stage_forward=None, conditioning_key=None, uncond_prob=0.2,
                 uncond_type="empty_seq", scale_factor=1.0, scale_by_std=False, encoder_type="2d", only_model=False,
                 noise_strength=0, use_dynamic_rescale=False, base_scale=0.7, turning_step=400, interp_mode=False,
                 fps_condition_type="fs", perframe_ae=False, logdir=None, rand_cond_frame=False,
                 en_and_decode_n_samples_a_time=None, *args, **kwargs):
        self.num_timesteps_cond = 1 if num_timesteps_cond is None else num_timesteps_cond
        self.scale_by_std = scale_by_std
        assert self.num_timesteps_cond <= kwargs["timesteps"]
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", [])
        super().__init__(conditioning_key=conditioning_key or "crossattn", *args, **kwargs)
        self.cond_stage_trainable = cond_stage_trainable
        self.cond_stage_key = cond_stage_key
        self.noise_strength = noise_strength
        self.use_dynamic_rescale = use_dynamic_rescale
        self.interp_mode = interp_mode
        self.fps_condition_type = fps_condition_type
        self.perframe_ae = perframe_ae
        # `perframe_ae` is the reference's memory relief (one AutoencoderKL call per frame, ddpm3d.py:633-639,657-663).
        # GroupNorm and the mid attention are per frame, so HOW MANY frames one launch sequence carries changes no
        # frame's arithmetic - only how many rows each kernel sees: at 4 frames per call the 72x128 .. 144x256 levels fill
        # the 256 CUs (16-frame clip @576x1024: encode 94 -> 74 ms, decode 161 -> 134 ms) at 6 GB of scratch.
        # The posterior noise is still drawn per frame, in frame order, from the CPU generator (seed parity).
        self.ae_frames_per_call = int(os.environ.get("DC_AE_FRAMES_PER_CALL", "4"))
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        if scale_by_std:
            self.register_buffer("scale_factor", torch.tensor(scale_factor))
        else:
            self.scale_factor = scale_factor
        if use_dynamic_rescale:
            # concat(linspace(1, base, turning_step), full(num_timesteps, base)): length 1400, as the reference
            arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(self.num_timesteps, base_scale)))
            self.register_buffer("scale_arr", torch.tensor(arr, dtype=torch.float32))
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        self.cond_stage_model = instantiate_from_config(cond_stage_config)
        if self.cond_stage_model is not None:
            self.cond_stage_model.eval()
        for p in self.parameters():
            p.requires_grad = False
        self.first_stage_config, self.cond_stage_config = first_stage_config, cond_stage_config
        self.clip_denoised = False
        self.cond_stage_forward = cond_stage_forward
        assert encoder_type in ("2d", "3d")
        self.encoder_type = encoder_type
        self.uncond_prob = uncond_prob
        self.classifier_free_guidance = uncond_prob > 0
        assert uncond_type in ("zero_embed", "empty_seq")
        self.uncond_type = uncond_type
        self.restarted_from_ckpt = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys, only_model=only_model)
            self.restarted_from_ckpt = True

    def get_learned_conditioning(self, c):
        m = self.cond_stage_model
        if self.cond_stage_forward is not None:
            return getattr(m, self.cond_stage_forward)(c)
        if hasattr(m, "encode") and callable(m.encode):
            c = m.encode(c)
            return c.mode() if isinstance(c, DiagonalGaussianDistribution) else c
        return m(c)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample(noise=noise)
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    @torch.no_grad()
    def encode_first_stage(self, x):
        """[b,c,t,h,w] (or [n,c,h,w]) pixels -> scaled latents; per-frame when `perframe_ae` (ddpm3d.py:620-644)."""
        five = self.encoder_type == "2d" and x.dim() == 5
        if five:
            b, _, t = x.shape[:3]
            x = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])
        if not self.perframe_ae:
            z = self.get_first_stage_encoding(self.first_stage_model.encode(x))
        else:
            k = max(1, self.ae_frames_per_call)
            fm = self.first_stage_model
            # every call's launch sequence first (asynchronous; each posterior owns a copy of its moments) ...
            posts = [fm.encode(x[i:i + k]) for i in range(0, x.shape[0], k)]
            noise_all = None
            if hasattr(fm, "latent_hw"):
                # ... then the posterior noise of every frame, while the GPU works: one CPU draw per frame in frame order, exactly
                # the draws the reference's per-frame posterior.sample() calls make (nothing else consumes the CPU
                # generator in between), crossing to the device once: a host draw + blocking copy between the
                # launch sequences leaves the GPU idle for as long as the host takes (measured 76 -> 165 ms per clip on
                # a busy box; drawn ahead of the first launch it still cost the clip the draw's own time)
                zh, zw = fm.latent_hw(x.shape[2], x.shape[3])
                noise_all = torch.cat([torch.randn((1, fm.embed_dim, zh, zw)) for _ in range(x.shape[0])], dim=0)
                noise_all = noise_all.to(x.device)
            outs = [self.get_first_stage_encoding(post, noise=None if noise_all is None else noise_all[j * k:j * k + k])
                    for j, post in enumerate(posts)]
            z = torch.cat(outs, dim=0)
        if five:
            z = z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()
        return z

    def decode_core(self, z, **kwargs):
        five = self.encoder_type == "2d" and z.dim() == 5
        if five:
            b, _, t = z.shape[:3]
            z = z.permute(0, 2, 1, 3, 4).reshape(b * t, z.shape[1], z.shape[3], z.shape[4])
        inv = 1. / self.scale_factor
        if not self.perframe_ae:
            out = self.first_stage_model.decode(inv * z, **kwargs)
        else:
            k = max(1, self.ae_frames_per_call)
            out = torch.cat([self.first_stage_model.decode(inv * z[i:i + k], **kwargs) for i in range(0, z.shape[0], k)], 0)
        if five:
            out = out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()
        return out

    @torch.no_grad()
    def decode_first_stage(self, z, **kwargs):
        return self.decode_core(z, **kwargs)

    def apply_model(self, x_noisy, t, cond, **kwargs):
        """ddpm3d.py:723-738"""
        if not isinstance(cond, dict):
            cond = cond if isinstance(cond, list) else [cond]
            cond = {"c_concat" if self.model.conditioning_key == "concat" else "c_crossattn": cond}
        kwargs = {k: v for k, v in kwargs.items() if k == "fs"}     # the UNet ignores the other pass-through kwargs
        out = self.model(x_noisy, t, **cond, **kwargs)
        return out[0] if isinstance(out, tuple) else out

    # ------------------------------------------------------------------ MI355X fast path
    def prepare_branches(self, x_shape, branches, fs=None):
        """Step-invariant part of a guided UNet evaluation: per-frame context rows of every conditioning branch
        (cond / uncond [/ image-only]), their c_concat tensors and the fs table. Done once per sampler call,
        outside the per-step (graph-captured) region; allocates."""
        net = self.model.diffusion_model
        key = self.model.conditioning_key
        B, Cx, T, H, W = x_shape
        dev = self.device
        nb = len(branches)
        ccs, ctx_all, Lc = [], None, None
        for k, cond in enumerate(branches):
            cc = None
            if key in ("hybrid", "concat"):
                cc = cond["c_concat"]
                cc = (cc[0] if len(cc) == 1 else torch.cat(cc, 1)).to(device=dev, dtype=torch.float32).contiguous()
            ccs.append(cc)
            ca = cond["c_crossattn"]
            ctx = (ca[0] if len(ca) == 1 else torch.cat(ca, 1)).to(dev)
            rows, Lc = net.build_context_rows(ctx, B, T, tag=f"ctx_b{k}")
            if ctx_all is None:
                ctx_all = net._arena.get("ctx_all", nb * rows.shape[0], rows.shape[1], device=dev)
            ops.copy2d(rows, ctx_all[k * rows.shape[0]:(k + 1) * rows.shape[0]])
        fs_table = None
        if net.fs_condition:
            if fs is None:
                fs = torch.full((B,), net.default_fs, dtype=torch.int64, device=dev)
            fs_table = fs.to(device=dev, dtype=torch.int64).repeat(nb).contiguous()
        # branches that share the latent AND the concat conditioning have identical activations up to the first
        # cross-attention: the UNet computes that prefix once (openaimodel3d.forward_rows shared_prefix)
        same_cc = all((c is ccs[0]) or (c is not None and ccs[0] is not None and c.shape == ccs[0].shape
                                         and bool(torch.equal(c, ccs[0]))) for c in ccs)
        share = nb if (same_cc and nb > 1 and os.environ.get("DC_SHARED_PREFIX", "1") != "0") else 1
        # the cross-attention K/V projections of the context are step-invariant too (attention.py:128-136): computed here,
        # once per sampler call, instead of in every UNet forward of every step
        ctx_kv = net.precompute_context_kv(ctx_all) if os.environ.get("DC_HOIST_CTX_KV", "1") != "0" else None
        return dict(nb=nb, ccs=ccs, ctx_all=ctx_all, Lc=Lc, fs_table=fs_table, shape=tuple(x_shape), share=share,
                    ctx_kv=ctx_kv)

    def apply_model_rows(self, x, prep, t_table, t_index=None):
        """All branches of `prep` on the same latent x as ONE batched UNet forward (kernel launches only; safe
        inside hipGraph capture). x fp32 contiguous [B,4,T,h,w]; t_table int64 [n_steps, nb*B] on device (row
        selected by the device counter t_index, else row 0).
        Returns fp32 rows [nb*B*T*h*w, 4] (branch-major): the layout dc_ddim_step consumes."""
        net = self.model.diffusion_model
        B, Cx, T, H, W = prep["shape"]
        nb = prep["nb"]
        M = B * T * H * W
        xr = net._arena.get("x_rows", nb * M, C_IN_PAD, device=x.device)
        for k, cc in enumerate(prep["ccs"]):
            ops.pack_latent(x, cc, xr[k * M:(k + 1) * M], B=B, Cx=Cx, Cc=0 if cc is None else cc.shape[1], T=T, HW=H * W)
        Lc = prep["Lc"]
        return net.forward_rows(xr, t_table, prep["ctx_all"], B=nb * B, T=T, H=H, W=W, Lc=Lc, n_text=min(77, Lc),
                                fs_table=prep["fs_table"], t_index=t_index, shared_prefix=prep.get("share", 1),
                                ctx_kv=prep.get("ctx_kv"))


class LatentVisualDiffusion(LatentDiffusion):
    def __init__(self, img_cond_stage_config, image_proj_stage_config, freeze_embedder=True,
                 image_proj_model_trainable=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.image_proj_model_trainable = image_proj_model_trainable
        self.embedder = instantiate_from_config(img_cond_stage_config)
        self.image_proj_model = instantiate_from_config(image_proj_stage_config)
        for m in (self.embedder, self.image_proj_model):
            if m is not None:
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
