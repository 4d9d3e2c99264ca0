"""CPU fp32 restatement of AutoencoderKL encode/decode (TEST INFRASTRUCTURE — see oracle/__init__.py).

Reference: lvdm/models/autoencoder.py:97-107, lvdm/modules/networks/ae_modules.py (Encoder :364-463,
Decoder :466-578, ResnetBlock :151-210, AttnBlock :26-78, Downsample :90-109, Upsample :111-127),
lvdm/distributions.py:24-65. Weights: flat dict keyed as `first_stage_model.*` minus that prefix.
"""
from dataclasses import dataclass
from typing import Sequence

import torch
import torch.nn.functional as F


@dataclass
class AECfg:
    ch: int = 128
    ch_mult: Sequence[int] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    in_channels: int = 3
    out_ch: int = 3
    z_channels: int = 4
    embed_dim: int = 4
    double_z: bool = True

    @staticmethod
    def from_params(ddconfig, embed_dim=4):
        known = {f for f in AECfg.__dataclass_fields__}
        kw = {k: v for k, v in ddconfig.items() if k in known}
        assert not ddconfig.get("attn_resolutions"), "attention only in the mid block for the released configs"
        return AECfg(embed_dim=embed_dim, **kw)


# the 576x1024 frame has 9216 mid-block positions: the score matrix is walked in query chunks above this many bytes
ATTN_CHUNK_BYTES = 1 << 28
ATTN_LARGE_IMPL = "sdpa"


def _swish(x):
    return x * torch.sigmoid(x)          # ae_modules.py:10-12


def _norm(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)   # ae_modules.py:15-16


def _conv(sd, p, x, **kw):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], **kw)


def resnet_block(sd, p, x):
    """ResnetBlock.forward ae_modules.py:190-210 (temb is None, dropout inactive)."""
    h = _conv(sd, p + ".conv1", _swish(_norm(sd, p + ".norm1", x)), padding=1)
    h = _conv(sd, p + ".conv2", _swish(_norm(sd, p + ".norm2", h)), padding=1)
    if p + ".nin_shortcut.weight" in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd, p, x):
    """AttnBlock.forward ae_modules.py:53-78: single head over h*w positions, scale c^-0.5."""
    h_ = _norm(sd, p + ".norm", x)
    q, k, v = _conv(sd, p + ".q", h_), _conv(sd, p + ".k", h_), _conv(sd, p + ".v", h_)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    v = v.reshape(b, c, h * w)
    n = h * w
    if 4 * n * n * b > ATTN_CHUNK_BYTES and ATTN_LARGE_IMPL == "sdpa":
        # fused online-softmax path for the 9216-position production frame (see oracle/unet.py:ATTN_LARGE_IMPL)
        o = F.scaled_dot_product_attention(q[:, None], k.permute(0, 2, 1)[:, None], v.permute(0, 2, 1)[:, None],
                                           scale=int(c) ** (-0.5))
        out = o[:, 0].permute(0, 2, 1)
    else:
        rows = max(1, min(n, ATTN_CHUNK_BYTES // (4 * n * b)))     # query rows per chunk (rows are independent: exact)
        out = torch.empty(b, c, n, dtype=x.dtype)
        for q0 in range(0, n, rows):
            w_ = torch.bmm(q[:, q0:q0 + rows], k) * (int(c) ** (-0.5))
            w_ = F.softmax(w_, dim=2)
            out[:, :, q0:q0 + rows] = torch.bmm(v, w_.permute(0, 2, 1))
    h_ = out.reshape(b, c, h, w)
    return x + _conv(sd, p + ".proj_out", h_)


@torch.no_grad()
def encoder_forward(sd, cfg, x, p="encoder"):
    """Encoder.forward ae_modules.py:430-463"""
    nres = len(cfg.ch_mult)
    h = _conv(sd, p + ".conv_in", x, padding=1)
    for lvl in range(nres):
        for i in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"{p}.down.{lvl}.block.{i}", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)       # Downsample :102-106
            h = _conv(sd, f"{p}.down.{lvl}.downsample.conv", h, stride=2)
    h = resnet_block(sd, p + ".mid.block_1", h)
    h = attn_block(sd, p + ".mid.attn_1", h)
    h = resnet_block(sd, p + ".mid.block_2", h)
    h = _swish(_norm(sd, p + ".norm_out", h))
    return _conv(sd, p + ".conv_out", h, padding=1)


@torch.no_grad()
def decoder_forward(sd, cfg, z, p="decoder"):
    """Decoder.forward ae_modules.py:539-578"""
    nres = len(cfg.ch_mult)
    h = _conv(sd, p + ".conv_in", z, padding=1)
    h = resnet_block(sd, p + ".mid.block_1", h)
    h = attn_block(sd, p + ".mid.attn_1", h)
    h = resnet_block(sd, p + ".mid.block_2", h)
    for lvl in reversed(range(nres)):
        for i in range(cfg.num_res_blocks + 1):
            h = resnet_block(sd, f"{p}.up.{lvl}.block.{i}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")       # Upsample :123-127
            h = _conv(sd, f"{p}.up.{lvl}.upsample.conv", h, padding=1)
    h = _swish(_norm(sd, p + ".norm_out", h))
    return _conv(sd, p + ".conv_out", h, padding=1)


@torch.no_grad()
def encode_moments(sd, cfg, x):
    """AutoencoderKL.encode autoencoder.py:97-102 -> moments [N, 2*embed, h, w]"""
    return _conv(sd, "quant_conv", encoder_forward(sd, cfg, x))


def posterior_sample(moments, noise=None):
    """DiagonalGaussianDistribution distributions.py:25-40,64 (noise None -> mode())."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * logvar) * noise


@torch.no_grad()
def decode(sd, cfg, z):
    """AutoencoderKL.decode autoencoder.py:104-107"""
    return decoder_forward(sd, cfg, _conv(sd, "post_quant_conv", z))


def encode_first_stage(sd, cfg, x, scale_factor, noise=None):
    """LatentDiffusion.encode_first_stage ddpm3d.py:620-644 + get_first_stage_encoding :611-618.
    x [b, c, t, h, w] -> z [b, zc, t, h/8, w/8]; per-frame or batched gives the same values."""
    b, c, t, h, w = x.shape
    frames = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    z = scale_factor * posterior_sample(encode_moments(sd, cfg, frames), noise)
    return z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)


def decode_first_stage(sd, cfg, z, scale_factor):
    """LatentDiffusion.decode_core ddpm3d.py:646-667"""
    b, c, t, h, w = z.shape
    frames = (1.0 / scale_factor * z).permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    out = decode(sd, cfg, frames)
    return out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4)


def ae_param_shapes(cfg):
    shapes = {}
    def norm(p, c):
        shapes[p + ".weight"] = (c,); shapes[p + ".bias"] = (c,)
    def conv(p, i, o, k):
        shapes[p + ".weight"] = (o, i, k, k); shapes[p + ".bias"] = (o,)
    def res(p, i, o):
        norm(p + ".norm1", i); conv(p + ".conv1", i, o, 3); norm(p + ".norm2", o); conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".nin_shortcut", i, o, 1)
    def attn(p, c):
        norm(p + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{p}.{n}", c, c, 1)
    nres = len(cfg.ch_mult)
    in_mult = (1,) + tuple(cfg.ch_mult)
    conv("encoder.conv_in", cfg.in_channels, cfg.ch, 3)
    bi = cfg.ch
    for lvl in range(nres):
        bi = cfg.ch * in_mult[lvl]; bo = cfg.ch * cfg.ch_mult[lvl]
        for i in range(cfg.num_res_blocks):
            res(f"encoder.down.{lvl}.block.{i}", bi, bo); bi = bo
        if lvl != nres - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", bi, bi, 3)
    res("encoder.mid.block_1", bi, bi); attn("encoder.mid.attn_1", bi); res("encoder.mid.block_2", bi, bi)
    norm("encoder.norm_out", bi)
    conv("encoder.conv_out", bi, 2 * cfg.z_channels if cfg.double_z else cfg.z_channels, 3)
    bi = cfg.ch * cfg.ch_mult[-1]
    conv("decoder.conv_in", cfg.z_channels, bi, 3)
    res("decoder.mid.block_1", bi, bi); attn("decoder.mid.attn_1", bi); res("decoder.mid.block_2", bi, bi)
    for lvl in reversed(range(nres)):
        bo = cfg.ch * cfg.ch_mult[lvl]
        for i in range(cfg.num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{i}", bi, bo); bi = bo
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", bi, bi, 3)
    norm("decoder.norm_out", bi); conv("decoder.conv_out", bi, cfg.out_ch, 3)
    conv("quant_conv", 2 * cfg.z_channels, 2 * cfg.embed_dim, 1)
    conv("post_quant_conv", cfg.embed_dim, cfg.z_channels, 1)
    return shapes
