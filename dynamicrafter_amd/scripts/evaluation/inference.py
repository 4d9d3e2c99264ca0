"""Host harness of the hot path: the counterpart of the reference's `scripts/evaluation/inference.py` functions that
sit between the data loader and the sampler.

    image_guided_synthesis   inference.py:216-313   conditioning assembly -> DDIM loop -> first-stage decode
    get_latent_z             inference.py:164-169   video -> per-frame AE latents
    load_model_checkpoint    inference.py:34-59     Lightning / DeepSpeed state dict -> model (key renames kept)

Same names, argument meaning and return layout as the reference, so `run_inference` style drivers can call them
unchanged. The model is this package's `LatentVisualDiffusion` on a HIP device; `model.embedder`,
`model.cond_stage_model` and `model.image_proj_model` are whatever the config instantiated (the OpenCLIP towers are
outside this package: see lvdm/modules/encoders/condition.py).
"""
from collections import OrderedDict

import torch

from ...lvdm.models.samplers.ddim import DDIMSampler
from ...lvdm.models.samplers.ddim_multiplecond import DDIMSampler as DDIMSampler_multicond


def load_model_checkpoint(model, ckpt):
    """inference.py:34-59. `ckpt` is a path (torch.load) or an already loaded mapping."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        if str(ckpt).endswith(".safetensors"):
            # flat tensor file with the same keys (the pruned community releases, reference README.md:384-385)
            from safetensors.torch import load_file
            state_dict = {"state_dict": load_file(str(ckpt), device="cpu")}
        else:
            state_dict = torch.load(ckpt, map_location="cpu")
    else:
        state_dict = ckpt
    if "state_dict" in state_dict:
        state_dict = state_dict["state_dict"]
        try:
            model.load_state_dict(state_dict, strict=True)
        except RuntimeError:
            # the 256x256 release names the frame-stride embedding `framestride_embed` (inference.py:41-51)
            renamed = OrderedDict((k.replace("framestride_embed", "fps_embedding"), v) for k, v in state_dict.items())
            model.load_state_dict(renamed, strict=True)
    else:
        # DeepSpeed checkpoints: {"module": {"_forward_module.<key>": tensor}} (inference.py:52-57: key[16:])
        model.load_state_dict(OrderedDict((k[16:], v) for k, v in state_dict["module"].items()))
    return model


def get_latent_z(model, videos):
    """inference.py:164-169: [b,c,t,h,w] pixels in [-1,1] -> [b,4,t,h/8,w/8] scaled latents."""
    b, c, t, h, w = videos.shape
    x = videos.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    z = model.encode_first_stage(x)
    return z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()


@torch.no_grad()
def image_guided_synthesis(model, prompts, videos, noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.,
                           unconditional_guidance_scale=1.0, cfg_img=None, fs=None, text_input=False,
                           multiple_cond_cfg=False, loop=False, interp=False, timestep_spacing="uniform",
                           guidance_rescale=0.0, use_fixed_scheduler=False, **kwargs):
    """inference.py:216-313. Returns [batch, n_samples, c, t, h, w] decoded frames.

    `use_fixed_scheduler` is accepted and ignored: the fork's "fixed" sampler only patches sigma so that
    1 - a_prev - sigma^2 cannot go negative (inference.py:172-214); the step kernel here clamps that radicand at zero
    (csrc/elementwise.hip), which is the same guard at the point of use.
    Extra keyword arguments (e.g. `x_T`, `noises`, `use_graph`) are passed to `DDIMSampler.sample` as the reference
    passes its **kwargs."""
    ddim_sampler = DDIMSampler_multicond(model) if multiple_cond_cfg else DDIMSampler(model)
    ddim_sampler.make_schedule(ddim_num_steps=ddim_steps, ddim_discretize=timestep_spacing, ddim_eta=ddim_eta, verbose=False)

    batch_size = noise_shape[0]
    fs = torch.tensor([fs] * batch_size, dtype=torch.long, device=model.device)
    if not text_input:
        prompts = [""] * batch_size

    img = videos[:, :, 0]                                         # b c h w
    img_emb = model.image_proj_model(model.embedder(img))         # b (t l) c
    cond_emb = model.get_learned_conditioning(prompts)
    cond = {"c_crossattn": [torch.cat([cond_emb, img_emb], dim=1)]}
    hybrid = model.model.conditioning_key == "hybrid"
    if hybrid:
        z = get_latent_z(model, videos)                           # b c t h w
        if loop or interp:
            img_cat_cond = torch.zeros_like(z)
            img_cat_cond[:, :, 0] = z[:, :, 0]
            img_cat_cond[:, :, -1] = z[:, :, -1]
        else:
            img_cat_cond = z[:, :, :1].repeat(1, 1, z.shape[2], 1, 1)
        cond["c_concat"] = [img_cat_cond]

    if unconditional_guidance_scale != 1.0:
        if model.uncond_type == "empty_seq":
            uc_emb = model.get_learned_conditioning(batch_size * [""])
        elif model.uncond_type == "zero_embed":
            uc_emb = torch.zeros_like(cond_emb)
        uc_img_emb = model.image_proj_model(model.embedder(torch.zeros_like(img)))
        uc = {"c_crossattn": [torch.cat([uc_emb, uc_img_emb], dim=1)]}
        if hybrid:
            uc["c_concat"] = [img_cat_cond]
    else:
        uc = None

    # the third branch: image yes, text "" (inference.py:266-273)
    if multiple_cond_cfg and cfg_img != 1.0:
        uc_2 = {"c_crossattn": [torch.cat([uc_emb, img_emb], dim=1)]}
        if hybrid:
            uc_2["c_concat"] = [img_cat_cond]
        kwargs.update({"unconditional_conditioning_img_nonetext": uc_2})
    else:
        kwargs.update({"unconditional_conditioning_img_nonetext": None})

    batch_variants = []
    for _ in range(n_samples):
        samples, _ = ddim_sampler.sample(S=ddim_steps, conditioning=cond, batch_size=batch_size, shape=noise_shape[1:],
                                         verbose=False, unconditional_guidance_scale=unconditional_guidance_scale,
                                         unconditional_conditioning=uc, eta=ddim_eta, cfg_img=cfg_img, mask=None,
                                         x0=None, fs=fs, timestep_spacing=timestep_spacing,
                                         guidance_rescale=guidance_rescale, **kwargs)
        batch_variants.append(model.decode_first_stage(samples))
    return torch.stack(batch_variants).permute(1, 0, 2, 3, 4, 5)
