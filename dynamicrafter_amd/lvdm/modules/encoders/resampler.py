"""Resampler — Perceiver-style projector from CLIP-vision tokens to per-frame image-context tokens, on the HIP path.

Drop-in for lvdm/modules/encoders/resampler.py:96-144 (Resampler), :48-93 (PerceiverAttention), :27-34
(FeedForward): same constructor keywords (`image_proj_stage_config.params`), same state_dict keys (`latents`,
`proj_in`, `proj_out`, `norm_out`, `layers.{i}.0.{norm1,norm2,to_q,to_kv,to_out}`, `layers.{i}.1.{0,1,3}`),
`forward(x [B, n1, embedding_dim]) -> [B, num_queries*video_length, output_dim]` (fp32).
Runs once per clip; it reuses dc_gemm_conv (GELU epilogue), dc_layernorm and
dc_flash_attn_d64 (dim_head must be 64, as in every released config).
"""
import torch
import torch.nn as nn

from .... import ops
from ....ops import PackedWeight
from ....param_tree import attach_params

_BF16 = torch.bfloat16


class Resampler(nn.Module):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, video_length=None):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("Resampler (HIP path): dim_head must be 64")
        if dim % 64 or embedding_dim % 64 or output_dim % 8:
            raise NotImplementedError("Resampler (HIP path): widths must be multiples of 64")
        self.num_queries = num_queries
        self.video_length = video_length
        self.dim, self.depth, self.heads = dim, depth, heads
        self.embedding_dim, self.output_dim = embedding_dim, output_dim
        nq = num_queries * video_length if video_length is not None else num_queries
        inner = dim_head * heads
        ffd = int(dim * ff_mult)
        t = {"latents": (1, nq, dim), "proj_in.weight": (dim, embedding_dim), "proj_in.bias": (dim,),
             "proj_out.weight": (output_dim, dim), "proj_out.bias": (output_dim,),
             "norm_out.weight": (output_dim,), "norm_out.bias": (output_dim,)}
        for i in range(depth):
            p = f"layers.{i}"
            t.update({f"{p}.0.norm1.weight": (dim,), f"{p}.0.norm1.bias": (dim,),
                      f"{p}.0.norm2.weight": (dim,), f"{p}.0.norm2.bias": (dim,),
                      f"{p}.0.to_q.weight": (inner, dim), f"{p}.0.to_kv.weight": (2 * inner, dim),
                      f"{p}.0.to_out.weight": (dim, inner),
                      f"{p}.1.0.weight": (dim,), f"{p}.1.0.bias": (dim,),
                      f"{p}.1.1.weight": (ffd, dim), f"{p}.1.3.weight": (dim, ffd)})
        attach_params(self, t)
        with torch.no_grad():
            self.latents.copy_(torch.randn(1, nq, dim, device=self.latents.device) / dim ** 0.5)
        self._packed = None
        self._arena = ops.Arena()
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_packed", None))

    def _p(self, name):
        node = self
        parts = name.split(".")
        for s in parts[:-1]:
            node = node._modules[s]
        return node._parameters[parts[-1]]

    def _buf(self, tag, rows, cols, dtype=_BF16, device=None):
        return self._arena.get(tag, rows, cols, dtype, device)

    def packed(self, device):
        if self._packed is not None and self._packed["device"] == device:
            return self._packed
        f32 = lambda n: self._p(n).detach().to(device=device, dtype=torch.float32).contiguous()
        lin = lambda n, bias=True: PackedWeight.linear(self._p(n + ".weight"), self._p(n + ".bias") if bias else None, device)
        P = {"device": device, "proj_in": lin("proj_in"), "proj_out": lin("proj_out"),
             "norm_out": (f32("norm_out.weight"), f32("norm_out.bias")),
             "latents": self.latents.detach()[0].to(device=device, dtype=_BF16).contiguous(), "layers": []}
        for i in range(self.depth):
            p = f"layers.{i}"
            P["layers"].append({"n1": (f32(p + ".0.norm1.weight"), f32(p + ".0.norm1.bias")),
                                "n2": (f32(p + ".0.norm2.weight"), f32(p + ".0.norm2.bias")),
                                "q": lin(p + ".0.to_q", False), "kv": lin(p + ".0.to_kv", False),
                                "o": lin(p + ".0.to_out", False),
                                "ffn": (f32(p + ".1.0.weight"), f32(p + ".1.0.bias")),
                                "ff1": lin(p + ".1.1", False), "ff2": lin(p + ".1.3", False)})
        self._packed = P
        return P

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("Resampler runs on the HIP path only (no CPU fallback)")
        dev = x.device
        P = self.packed(dev)
        B, n1, E = x.shape
        L = P["latents"].shape[0]
        D, inner = self.dim, self.heads * 64
        xin = self._buf("x_in", B * n1, E, device=dev)
        ops.nchw_to_rows(x.to(torch.float32).contiguous(), xin, N=B * n1, Cc=E, HW=1)      # fp32 -> bf16 rows
        xr = ops.gemm(xin, P["proj_in"], self._buf("x", B * n1, D, device=dev))
        lat = self._buf("lat", B * L, D, device=dev)
        for b in range(B):
            ops.copy2d(P["latents"], lat[b * L:(b + 1) * L])                                # latents.repeat(B,1,1)
        for W in P["layers"]:
            xn = ops.layernorm(xr, self._buf("xn", B * n1, D, device=dev), *W["n1"])
            ln = ops.layernorm(lat, self._buf("ln", B * L, D, device=dev), *W["n2"])
            q = ops.gemm(ln, W["q"], self._buf("q", B * L, inner, device=dev))
            kv = self._buf("kv", B * (n1 + L), 2 * inner, device=dev)                       # cat((x, latents), dim=-2)
            for b in range(B):
                base = b * (n1 + L)
                ops.gemm(xn[b * n1:(b + 1) * n1], W["kv"], kv[base:base + n1])
                ops.gemm(ln[b * L:(b + 1) * L], W["kv"], kv[base + n1:base + n1 + L])
            att = self._buf("att", B * L, inner, device=dev)
            # (q*s)(k*s)^T with s = d^-1/4  ==  q k^T * d^-1/2
            ops.flash_attn(q, kv[:, :inner], kv[:, inner:], att, batch=B, heads=self.heads, Lq=L, Lk=n1 + L, scale=0.125)
            ops.gemm(att, W["o"], lat, residual=lat)
            n = ops.layernorm(lat, self._buf("ln", B * L, D, device=dev), *W["ffn"])
            h = ops.gemm(n, W["ff1"], self._buf("ffh", B * L, W["ff1"].N, device=dev), gelu=True)
            ops.gemm(h, W["ff2"], lat, residual=lat)
        o = ops.gemm(lat, P["proj_out"], self._buf("o", B * L, self.output_dim, device=dev))
        on = ops.layernorm(o, self._buf("on", B * L, self.output_dim, device=dev), *P["norm_out"])
        out = torch.empty((B, L, self.output_dim), dtype=torch.float32, device=dev)
        ops.rows_to_nchw(on, out, N=B * L, Cc=self.output_dim, HW=1)
        return out
