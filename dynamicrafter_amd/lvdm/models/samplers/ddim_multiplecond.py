"""`lvdm.models.samplers.ddim_multiplecond.DDIMSampler` (reference ddim_multiplecond.py:11-323): the sampler with the
third guidance branch e_u + cfg_img (e_ui - e_u) + s (e_c - e_ui).

This package's DDIMSampler already evaluates that branch when `cfg_img` and
`unconditional_conditioning_img_nonetext` are passed (samplers/ddim.py), so this module only provides the reference's
import path; with neither argument it behaves as the two-branch sampler, exactly as the reference class does when
`unconditional_conditioning_img_nonetext` is None... except that the reference then raises on `e_t_uncond_img`
being undefined - callers never reach that case (inference.py:268-273 always sets the kwarg)."""
from .ddim import DDIMSampler as _DDIMSampler


class DDIMSampler(_DDIMSampler):
    pass
