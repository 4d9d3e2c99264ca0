"""CPU fp32 restatement of the Resampler (TEST INFRASTRUCTURE — see oracle/__init__.py).
Reference: lvdm/modules/encoders/resampler.py:96-144 (Resampler), :48-93 (PerceiverAttention), :27-34 (FeedForward)."""
import math

import torch
import torch.nn.functional as F


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def perceiver_attention(sd, p, x, latents, heads):
    """resampler.py:64-93"""
    x = _ln(sd, p + ".norm1", x)
    latents = _ln(sd, p + ".norm2", latents)
    b, l, _ = latents.shape
    q = F.linear(latents, sd[p + ".to_q.weight"])
    k, v = F.linear(torch.cat((x, latents), dim=-2), sd[p + ".to_kv.weight"]).chunk(2, dim=-1)
    def split(t):
        return t.reshape(b, t.shape[1], heads, -1).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    scale = 1 / math.sqrt(math.sqrt(q.shape[-1]))
    w = torch.softmax(((q * scale) @ (k * scale).transpose(-2, -1)).float(), dim=-1)
    out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
    return F.linear(out, sd[p + ".to_out.weight"])


@torch.no_grad()
def resampler_forward(sd, x, heads, depth):
    """resampler.py:132-144"""
    latents = sd["latents"].repeat(x.size(0), 1, 1)
    x = F.linear(x, sd["proj_in.weight"], sd["proj_in.bias"])
    for i in range(depth):
        latents = perceiver_attention(sd, f"layers.{i}.0", x, latents, heads) + latents
        h = _ln(sd, f"layers.{i}.1.0", latents)
        h = F.linear(F.gelu(F.linear(h, sd[f"layers.{i}.1.1.weight"])), sd[f"layers.{i}.1.3.weight"])
        latents = h + latents
    latents = F.linear(latents, sd["proj_out.weight"], sd["proj_out.bias"])
    return _ln(sd, "norm_out", latents)
