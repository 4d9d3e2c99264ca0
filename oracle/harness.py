"""CPU restatement of the inference harness between the data loader and the sampler (TEST INFRASTRUCTURE — see
oracle/__init__.py). Reference: scripts/evaluation/inference.py:216-313 (image_guided_synthesis), :164-169
(get_latent_z). The learned pieces are the other oracle modules; the two CLIP towers are callables supplied by the
caller (the tests use the deterministic stand-ins of tests/golden_cfg.py, as the fixture generator does)."""
import torch

from . import ddim as oddim
from . import resampler as ores
from . import unet as ounet
from . import vae as ovae


@torch.no_grad()
def image_guided_synthesis(*, unet_sd, unet_cfg, ae_sd, ae_cfg, proj_sd, proj_heads, proj_depth, embed_image, embed_text,
                           schedule: oddim.ModelSchedule, scale_factor, uncond_type, prompts, videos, x_T, noises,
                           ae_noise, ddim_steps=50, ddim_eta=1.0, unconditional_guidance_scale=1.0, cfg_img=None, fs=None,
                           text_input=False, multiple_cond_cfg=False, loop=False, interp=False,
                           timestep_spacing="uniform", guidance_rescale=0.0):
    """Returns [b, 1, c, t, h, w] (one sample per clip)."""
    b = videos.shape[0]
    fs_t = torch.tensor([fs] * b, dtype=torch.long)                                   # :233
    if not text_input:
        prompts = [""] * b                                                            # :235-236
    img = videos[:, :, 0]                                                             # :238
    project = lambda e: ores.resampler_forward(proj_sd, e, proj_heads, proj_depth)
    img_emb = project(embed_image(img))                                               # :239-240
    cond_emb = embed_text(prompts)                                                    # :242
    ctx = torch.cat([cond_emb, img_emb], dim=1)                                       # :243
    z = ovae.encode_first_stage(ae_sd, ae_cfg, videos, scale_factor, ae_noise)        # :245 get_latent_z
    if loop or interp:                                                                # :246-249
        cc = torch.zeros_like(z)
        cc[:, :, 0] = z[:, :, 0]
        cc[:, :, -1] = z[:, :, -1]
    else:                                                                             # :250-252
        cc = z[:, :, :1].repeat(1, 1, z.shape[2], 1, 1)
    uc_ctx = uc2_ctx = None
    if unconditional_guidance_scale != 1.0:                                           # :255-265
        uc_emb = embed_text([""] * b) if uncond_type == "empty_seq" else torch.zeros_like(cond_emb)
        uc_ctx = torch.cat([uc_emb, project(embed_image(torch.zeros_like(img)))], dim=1)
        if multiple_cond_cfg and cfg_img != 1.0:                                      # :268-273
            uc2_ctx = torch.cat([uc_emb, img_emb], dim=1)
    sched = oddim.DDIMSchedule(schedule, ddim_steps, timestep_spacing, ddim_eta)

    def apply_model(x, t, c, fs=None):                                                # DiffusionWrapper 'hybrid'
        return ounet.unet_forward(unet_sd, unet_cfg, torch.cat([x, cc], dim=1), t, c, fs)

    samples = oddim.ddim_sample(apply_model, sched, x_T, ctx, uc_ctx, cfg_scale=unconditional_guidance_scale,
                                guidance_rescale=guidance_rescale, noises=None if ddim_eta == 0 else list(noises),
                                uncond_img=uc2_ctx, cfg_img=cfg_img, fs=fs_t)
    out = ovae.decode_first_stage(ae_sd, ae_cfg, samples, scale_factor)               # :309
    return out[:, None]                                                               # :311-313 (n_samples = 1)
