"""DiagonalGaussianDistribution over VAE moments kept as channels-last device rows.

Drop-in for lvdm/distributions.py:24-65 on the inference path: `sample(noise=None)`, `mode()`, and the
`parameters / mean / logvar / std / var` views. `sample()` without noise draws `torch.randn(shape)` from the CPU
default generator exactly like the reference (:37) so that a seeded script consumes the same random stream.
"""
import torch

from .. import ops


class DiagonalGaussianDistribution:
    def __init__(self, moment_rows, zc, N, H, W, deterministic=False):
        self._rows = moment_rows          # bf16 [N*H*W, >= 2*zc]: mean | logvar
        self._zc, self._N, self._H, self._W = zc, N, H, W
        self.deterministic = deterministic

    def _shape(self):
        return (self._N, self._zc, self._H, self._W)

    def sample(self, noise=None):
        dev = self._rows.device
        if self.deterministic:
            return self.mode()
        if noise is None:
            noise = torch.randn(self._shape())
        # the kernel reads one value per output element: broadcastable noise (the reference computes
        # mean + std * noise with torch broadcasting) is expanded here, anything else is refused before the launch
        try:
            noise = noise.to(device=dev, dtype=torch.float32).expand(self._shape()).contiguous()
        except RuntimeError as e:
            raise ValueError(f"noise of shape {tuple(noise.shape)} does not broadcast to {self._shape()}") from e
        z = torch.empty(self._shape(), dtype=torch.float32, device=dev)
        return ops.vae_sample(self._rows, noise, z, N=self._N, zc=self._zc, HW=self._H * self._W, scale=1.0)

    def mode(self):
        z = torch.empty(self._shape(), dtype=torch.float32, device=self._rows.device)
        return ops.vae_sample(self._rows, None, z, N=self._N, zc=self._zc, HW=self._H * self._W, scale=1.0)

    @property
    def parameters(self):
        out = torch.empty((self._N, 2 * self._zc, self._H, self._W), dtype=torch.float32, device=self._rows.device)
        return ops.rows_to_nchw(self._rows, out, N=self._N, Cc=2 * self._zc, HW=self._H * self._W)

    @property
    def mean(self):
        return self.parameters[:, :self._zc]

    @property
    def logvar(self):
        return torch.clamp(self.parameters[:, self._zc:], -30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        return torch.exp(self.logvar)
