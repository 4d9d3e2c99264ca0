"""Host-side (NumPy fp64 / torch fp32) diffusion schedule tables — tiny, computed once per sampler call.

Same function names and results as the reference's lvdm/models/utils_diffusion.py (:31-157); the numbers are
pinned bit-exactly to the reference by tests/golden/schedules.npz. Only what inference needs is provided.
"""
import numpy as np
import torch


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule != "linear":
        raise NotImplementedError(f"beta schedule '{schedule}' (the released configs use 'linear')")
    ramp = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64, device="cpu")
    return (ramp * ramp).numpy()


def rescale_zero_terminal_snr(betas):
    """Zero-terminal-SNR rescale (arXiv 2305.08891 alg. 1): shift sqrt(abar) so abar_T = 0, keep abar_0."""
    root = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    first, last = root[0].copy(), root[-1].copy()
    root -= last
    root *= first / (first - last)
    abar = root ** 2
    step = np.concatenate([abar[:1], abar[1:] / abar[:-1]])
    return 1 - step


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.asarray(list(range(0, num_ddpm_timesteps, stride))) + 1
    elif ddim_discr_method == "uniform_trailing":
        stride = num_ddpm_timesteps / num_ddim_timesteps
        steps = np.flip(np.round(np.arange(num_ddpm_timesteps, 0, -stride))).astype(np.int64) - 1
    elif ddim_discr_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int) + 1
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps}")
    return steps


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """Returns (sigmas fp64, alphas fp32 tensor, alphas_prev fp64 ndarray) with the reference's dtype path:
    alphacums is the model's fp32 buffer; 1/(1-alphas) is taken in fp32 (torch's reflected division) before
    the promotion to fp64 — see oracle/ddim.py for the probe that established this."""
    alphacums = torch.as_tensor(alphacums)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0].item()] + alphacums[ddim_timesteps[:-1]].tolist())
    prev64 = torch.tensor(alphas_prev, dtype=torch.float64)
    spread = (1 - alphas).reciprocal().double() * (1 - prev64)
    sigmas = eta * torch.sqrt(spread * (1 - alphas.double() / prev64))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the sigma_t schedule {sigmas}")
    return sigmas, alphas, alphas_prev
