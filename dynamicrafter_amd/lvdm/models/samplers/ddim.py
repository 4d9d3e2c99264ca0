"""DDIMSampler — the reference's sampler API (lvdm/models/samplers/ddim.py:10-317) over the fused HIP step.

`DDIMSampler(model).sample(S, batch_size, shape, conditioning, ..., x_T, unconditional_guidance_scale,
unconditional_conditioning, eta, fs, timestep_spacing, guidance_rescale, **kwargs) -> (samples, intermediates)`
behaves as the reference's, including the 3-branch guidance of ddim_multiplecond.py (pass `cfg_img` and
`unconditional_conditioning_img_nonetext`). Differences by design:
  * per step, the cond / uncond (/ image-only) UNet evaluations run as ONE batched forward
    (`model.apply_model_rows`) and CFG combine + guidance-rescale + v->eps/x0 + dynamic rescale + the DDIM update
    are one fused kernel (dc_ddim_step) instead of ~25 elementwise launches and 6 host->device scalars;
  * per-step scalars live in device tables indexed by a device-side step counter, so one captured hipGraph of
    a step is replayed S times (`use_graph=True`);
  * `noises=` (tensor [S, *x.shape]) injects the per-step Gaussian noise (parity tests); otherwise it is drawn
    with torch.randn on the device as the reference does (common.py:31-34).
  * sqrt(1 - a_prev - sigma^2) is clamped at 0 (the reference can produce NaN there: SURVEY §8 a2).
  * `mask` / `x0` blending (reference :174-180) is one more kernel ahead of the step (dc_mask_blend: q_sample + blend),
    inside the captured graph; `q_noises=` (tensor [S, *x.shape]) injects the q_sample draws (parity tests).
"""
import numpy as np
import torch

from .... import _hip, ops
from ..utils_diffusion import make_ddim_sampling_parameters, make_ddim_timesteps


class FusedRun:
    """State of one fused sampling run: the latent (updated in place), device tables / counter, the batched
    guidance branches, and optionally the captured hipGraph of a step. `step()` advances the device-side DDIM
    index by one; after `S` steps `rewind()` starts the next clip."""

    def __init__(self, sampler, img, branches, *, fs=None, noises=None, cfg_scale=1.0, cfg_img=None,
                 guidance_rescale=0.0, temperature=1.0, mask=None, x0=None, q_noises=None, clean_cond=False):
        m = sampler.model
        self.sampler, self.model = sampler, m
        self.img = img
        dev = img.device
        b = img.shape[0]
        self.S = sampler._exec_timesteps.shape[0]
        self.nb = len(branches)
        self.prep = m.prepare_branches(tuple(img.shape), branches, fs=fs)
        for name, t in (("noises", noises), ("q_noises", None if (mask is None or clean_cond) else q_noises)):
            if t is not None and t.numel() < self.S * img.numel():
                raise ValueError(f"{name}: the step kernels read step index * {img.numel()} + i for {self.S} steps; "
                                 f"got {t.numel()} elements")
        self.noises = noises
        self.pred_x0 = torch.empty_like(img)
        self.ws = torch.empty(16 * b * 256, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        t_host = torch.as_tensor(sampler._exec_timesteps.copy(), dtype=torch.int64)
        self.t_table = t_host[:, None].repeat(1, self.nb * b).contiguous().to(dev)          # [S, nb*B]
        self.kw = dict(B=b, Cc=img.shape[1], THW=int(np.prod(img.shape[2:])), v_param=m.parameterization == "v",
                       cfg_scale=cfg_scale, cfg_img=cfg_scale if cfg_img is None else cfg_img,
                       guidance_rescale=guidance_rescale, temperature=temperature, noise_step_stride=img.numel())
        self.graph = None
        # mask / x0 blending ahead of every step (ddim.py:174-180): the original latent, re-noised to the step's
        # timestep with pre-drawn q_sample noise [S, ...] unless clean_cond
        self.blend = None
        if mask is not None:
            assert x0 is not None
            full = lambda t: t.to(device=dev, dtype=torch.float32).expand_as(img).contiguous()
            if not clean_cond and q_noises is None:
                q_noises = torch.randn((self.S,) + tuple(img.shape), device=dev)
            self.blend = dict(x0=full(x0), mask=full(mask), clean=bool(clean_cond),
                              q=None if clean_cond else q_noises.to(device=dev, dtype=torch.float32).contiguous())

    def _enqueue(self):
        """Kernel launches of one step (no allocation, no host sync): [mask blend +] batched UNet + fused DDIM update."""
        if self.blend is not None:
            b = self.blend
            ops.mask_blend(self.img, b["x0"], b["mask"], b["q"], self.sampler._tables, step_index=self.counter,
                           clean=b["clean"], noise_step_stride=self.img.numel())
        e = self.model.apply_model_rows(self.img, self.prep, self.t_table, t_index=self.counter)
        M = self.kw["B"] * self.kw["THW"]
        e_u = e[M:2 * M] if self.nb > 1 else None
        e_i = e[2 * M:3 * M] if self.nb > 2 else None
        ops.ddim_step(self.sampler._tables, e[:M], e_u, e_i, self.img, self.noises, self.img, self.pred_x0, self.ws,
                      step_index=self.counter, **self.kw)
        ops.advance_counter(self.counter)

    def capture(self):
        """Warm up once eagerly (allocates all scratch), restore the state, capture one step into a hipGraph."""
        keep = self.img.clone()
        self._enqueue()
        torch.cuda.synchronize()
        self.img.copy_(keep)
        self.counter.zero_()
        torch.cuda.synchronize()
        self.graph = ops.DeviceGraph().capture(self._enqueue)
        return self

    def step(self):
        if self.graph is not None:
            self.graph.launch()
        else:
            self._enqueue()

    def rewind(self, x_T=None):
        self.sync()
        self.counter.zero_()
        if x_T is not None:
            self.img.copy_(x_T)
        torch.cuda.synchronize()

    def sync(self):
        """Wait for the issued steps, then read the library's error word: a kernel whose bounded LDS-counter wait timed out
        (csrc/gemm_pipe.h) has produced garbage, and that must not leave the sampler silently."""
        if self.graph is not None:
            self.graph.sync()
        else:
            torch.cuda.current_stream().synchronize()
        _hip.check_error_word("FusedRun.sync")


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0
        self._graph = None
        self._graph_key = None

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    # ------------------------------------------------------------------ schedule (host, once per call)
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        m = self.model
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        acp = m.alphas_cumprod.detach().float().cpu()
        assert acp.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        dev = m.device
        if getattr(m, "use_dynamic_rescale", False):
            sa = m.scale_arr.detach().cpu()[self.ddim_timesteps]
            self.ddim_scale_arr = sa
            self.ddim_scale_arr_prev = torch.cat([sa[0:1], sa[:-1]])
        sig, a, a_prev = make_ddim_sampling_parameters(alphacums=acp, ddim_timesteps=self.ddim_timesteps, eta=ddim_eta,
                                                       verbose=verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, a, a_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - a.numpy())
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod"):
            setattr(self, k, getattr(m, k).detach().float().to(dev))
        # fp32 tables in EXECUTION order (step i uses DDIM index S-1-i): what torch.full(size, v) would hold
        S = self.ddim_timesteps.shape[0]
        order = np.arange(S)[::-1].copy()
        f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float64)).float()[order].contiguous().to(dev)
        ts = torch.as_tensor(self.ddim_timesteps.copy(), dtype=torch.long)
        t = {"a_t": a.float()[order].contiguous().to(dev), "a_prev": f32(a_prev), "sigma_t": f32(sig.numpy()),
             "sqrt_one_minus_at": torch.as_tensor(self.ddim_sqrt_one_minus_alphas)[order].contiguous().to(dev),
             "sqrt_acp_t": m.sqrt_alphas_cumprod.detach().float().cpu()[ts][order].contiguous().to(dev),
             "sqrt_1macp_t": m.sqrt_one_minus_alphas_cumprod.detach().float().cpu()[ts][order].contiguous().to(dev)}
        if getattr(m, "use_dynamic_rescale", False):
            t["scale_ratio"] = (self.ddim_scale_arr_prev / self.ddim_scale_arr)[order].contiguous().to(dev)
        self._tables = t
        self._exec_timesteps = np.flip(self.ddim_timesteps).copy()

    # ------------------------------------------------------------------ public API
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, schedule_verbose=False, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, precision=None, fs=None,
               timestep_spacing="uniform", guidance_rescale=0.0, **kwargs):
        if conditioning is not None and isinstance(conditioning, dict):
            first = conditioning[list(conditioning.keys())[0]]
            cbs = (first[0] if isinstance(first, (list, tuple)) else first).shape[0]
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_discretize=timestep_spacing, ddim_eta=eta, verbose=schedule_verbose)
        size = (batch_size,) + tuple(shape)
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback, mask=mask, x0=x0,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose,
                                  precision=precision, fs=fs, guidance_rescale=guidance_rescale,
                                  quantize_denoised=quantize_x0, **kwargs)

    def _branches(self, cond, uc, scale, kwargs):
        br = [cond]
        if uc is not None and scale != 1.:
            br.append(uc)
            uc2 = kwargs.get("unconditional_conditioning_img_nonetext")
            if uc2 is not None:
                br.append(uc2)
        return br

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100, temperature=1.,
                      noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, verbose=True, precision=None, fs=None, guidance_rescale=0.0,
                      noises=None, use_graph=False, **kwargs):
        if ddim_use_original_steps or timesteps is not None or quantize_denoised or score_corrector is not None \
                or noise_dropout > 0.:
            raise NotImplementedError("only the options DynamiCrafter inference uses are implemented "
                                      "(no original-steps / partial / quantised / corrected sampling)")
        m = self.model
        dev = m.device
        if dev.type != "cuda":
            raise RuntimeError("DDIMSampler runs on the HIP path only: put the model on the GPU")
        b = shape[0]
        img = (torch.randn(shape, device=dev) if x_T is None else x_T.to(dev)).to(torch.float32).contiguous().clone()
        S = self._exec_timesteps.shape[0]
        clean_cond = kwargs.pop("clean_cond", False)
        cfg_img = kwargs.get("cfg_img")
        branches = self._branches(cond, unconditional_conditioning, unconditional_guidance_scale, kwargs)
        nb = len(branches)
        if cfg_img is None:
            cfg_img = unconditional_guidance_scale
        eta_on = bool((self._tables["sigma_t"] != 0).any().item())
        q_noises = kwargs.pop("q_noises", None)
        draw_q = mask is not None and not clean_cond and q_noises is None
        if noises is None or draw_q:
            # drawn up front so a captured graph can index them by the device step counter - in the order of the reference's
            # per-step draws: with a mask, q_sample's randn_like of step i comes before that step's noise_like
            # (ddim.py:174-180, then p_sample_ddim :270). The reference calls noise_like on EVERY step, also when sigma_t = 0
            # (eta = 0): the draw is then made and discarded here too, so the generator - and with a mask the q_sample noises
            # of the later steps - stays in step with the reference for the same seed.
            qs, ns = [], []
            for _ in range(S):
                if draw_q:
                    qs.append(torch.randn(shape, device=dev))
                if noises is None:
                    n_i = torch.randn(shape, device=dev)
                    if eta_on:
                        ns.append(n_i)
            if qs:
                q_noises = torch.stack(qs)
            if ns:
                noises = torch.stack(ns)
        if noises is not None:
            noises = noises.to(device=dev, dtype=torch.float32).contiguous()
        if noises is not None and noises.numel() < S * img.numel():
            raise ValueError(f"noises: {S} steps x {img.numel()} elements needed, got {noises.numel()}")
        if mask is not None and not clean_cond and q_noises.numel() < S * img.numel():
            raise ValueError(f"q_noises: {S} steps x {img.numel()} elements needed, got {q_noises.numel()}")
        fast = hasattr(m, "apply_model_rows") and all(isinstance(c, dict) for c in branches)
        intermediates = {"x_inter": [img.clone()], "pred_x0": [img.clone()]}

        if fast:
            run = FusedRun(self, img, branches, fs=fs, noises=noises, cfg_scale=unconditional_guidance_scale,
                           cfg_img=cfg_img, guidance_rescale=guidance_rescale, temperature=temperature, mask=mask,
                           x0=x0, q_noises=q_noises, clean_cond=clean_cond)
            if use_graph:
                run.capture()
            for i in range(S):
                run.step()
                index = S - i - 1
                log_now = index % log_every_t == 0 or index == S - 1
                if use_graph and (callback or img_callback or log_now):
                    run.sync()                 # the graph runs on its own stream: the step must have finished before the
                if callback: callback(i)       # callbacks / snapshots read its results (ddim.py:196-201)
                if img_callback: img_callback(run.pred_x0, i)
                if log_now:
                    intermediates["x_inter"].append(img.clone())
                    intermediates["pred_x0"].append(run.pred_x0.clone())
            run.sync()
            self._last_run = run
            return img, intermediates

        # generic path: any model exposing apply_model(x, t, c, **kw) -> [B, C, ...] (also mask / x0 blending)
        for i, step in enumerate(self._exec_timesteps):
            index = S - i - 1
            ts = torch.full((b,), int(step), device=dev, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                full = lambda t: t.to(device=dev, dtype=torch.float32).expand_as(img).contiguous()
                qn = None if clean_cond else q_noises[i].to(dev, torch.float32).contiguous()
                img = ops.mask_blend(img.contiguous().clone(), full(x0), full(mask), qn, self._tables, index=i,
                                     clean=clean_cond)
            img, pred_x0 = self.p_sample_ddim(img, cond, ts, index=index, temperature=temperature,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning, fs=fs,
                                              guidance_rescale=guidance_rescale,
                                              noise=None if noises is None else noises[i], **kwargs)
            if callback: callback(i)
            if img_callback: img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == S - 1:
                intermediates["x_inter"].append(img)
                intermediates["pred_x0"].append(pred_x0)
        return img, intermediates

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, uc_type=None,
                      conditional_guidance_scale_temporal=None, mask=None, x0=None, guidance_rescale=0.0, noise=None,
                      cfg_img=None, **kwargs):
        """One DDIM update from separate apply_model calls (reference :205-279 / multiplecond :211-285)."""
        if use_original_steps or quantize_denoised or score_corrector is not None or noise_dropout > 0.:
            raise NotImplementedError
        m = self.model
        dev = x.device
        x = x.to(torch.float32).contiguous()
        uc2 = kwargs.pop("unconditional_conditioning_img_nonetext", None)
        kwargs.pop("clean_cond", None)
        mk = {k: v for k, v in kwargs.items() if k == "fs"}
        e_c = m.apply_model(x, t, c, **mk).to(torch.float32).contiguous()
        e_u = e_i = None
        if unconditional_conditioning is not None and unconditional_guidance_scale != 1.:
            e_u = m.apply_model(x, t, unconditional_conditioning, **mk).to(torch.float32).contiguous()
            if uc2 is not None:
                e_i = m.apply_model(x, t, uc2, **mk).to(torch.float32).contiguous()
        if cfg_img is None:
            cfg_img = unconditional_guidance_scale
        S = self._exec_timesteps.shape[0]
        if noise is None:
            noise = torch.randn(x.shape, device=dev)        # drawn even when sigma == 0, as the reference does
            if repeat_noise:
                noise = noise[:1].expand_as(x)
        noise = noise.to(torch.float32).contiguous()
        b = x.shape[0]
        x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
        ws = torch.empty(16 * b * 256, dtype=torch.float32, device=dev)
        ops.ddim_step(self._tables, e_c, e_u, e_i, x, noise, x_prev, pred_x0, ws, B=b, Cc=x.shape[1],
                      THW=int(np.prod(x.shape[2:])), index=S - 1 - index, v_param=m.parameterization == "v",
                      cfg_scale=unconditional_guidance_scale, cfg_img=cfg_img, guidance_rescale=guidance_rescale,
                      temperature=temperature, e_nchw=True)
        return x_prev, pred_x0

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """ddim.py:303-317 (tiny elementwise helper, torch)."""
        if use_original_steps:
            sa, s1 = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        else:
            sa = torch.sqrt(self.ddim_alphas).to(x0.device)
            s1 = torch.as_tensor(self.ddim_sqrt_one_minus_alphas).to(x0.device)
        noise = torch.randn_like(x0) if noise is None else noise
        shp = (t.shape[0],) + (1,) * (x0.dim() - 1)
        return sa.to(x0.device)[t].reshape(shp) * x0 + s1[t].reshape(shp) * noise

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, callback=None):
        """ddim.py:281-301: run the last t_start DDIM steps from x_latent."""
        if use_original_steps:
            raise NotImplementedError
        S = self.ddim_timesteps.shape[0]
        steps = np.flip(self.ddim_timesteps[:t_start])
        x_dec = x_latent
        for i, step in enumerate(steps):
            index = steps.shape[0] - i - 1
            ts = torch.full((x_latent.shape[0],), int(step), device=x_latent.device, dtype=torch.long)
            x_dec, _ = self.p_sample_ddim(x_dec, cond, ts, index=index,
                                          unconditional_guidance_scale=unconditional_guidance_scale,
                                          unconditional_conditioning=unconditional_conditioning)
            if callback: callback(i)
        return x_dec
