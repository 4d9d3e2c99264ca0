"""CPU fp32 restatement of the reference 3D UNet forward (TEST INFRASTRUCTURE — see oracle/__init__.py).

Functional: weights come in as a flat dict keyed by the reference's state_dict names (relative to
`model.diffusion_model.`), hyper-parameters as `UNetCfg` (the `unet_config.params` block of configs/*.yaml).
Reference: lvdm/modules/networks/openaimodel3d.py (UNetModel :281-603), lvdm/modules/attention.py.
"""
import math
from dataclasses import dataclass, field
from typing import Sequence

import torch
import torch.nn.functional as F


@dataclass
class UNetCfg:
    in_channels: int = 8
    out_channels: int = 4
    model_channels: int = 320
    attention_resolutions: Sequence[int] = (4, 2, 1)
    num_res_blocks: int = 2
    channel_mult: Sequence[int] = (1, 2, 4, 4)
    num_head_channels: int = 64
    context_dim: int = 1024
    temporal_length: int = 16
    temporal_conv: bool = True
    temporal_attention: bool = True
    addition_attention: bool = True
    image_cross_attention: bool = True
    image_cross_attention_scale_learnable: bool = False
    default_fs: int = 4
    fs_condition: bool = True
    use_linear: bool = True
    text_context_len: int = 77
    # accepted and ignored (inference): dropout, use_checkpoint, transformer_depth==1, ...
    extra: dict = field(default_factory=dict)

    @staticmethod
    def from_params(params):
        known = {f for f in UNetCfg.__dataclass_fields__ if f != "extra"}
        kw = {k: v for k, v in params.items() if k in known}
        cfg = UNetCfg(**kw)
        cfg.extra = {k: v for k, v in params.items() if k not in known}
        assert cfg.extra.get("transformer_depth", 1) == 1
        assert cfg.extra.get("temporal_selfatt_only", True)
        assert not cfg.extra.get("use_relative_position", False)
        assert not cfg.extra.get("use_causal_attention", False)
        return cfg


def timestep_embedding(t, dim, max_period=10000):
    """utils_diffusion.py:8-28"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _mlp(sd, p, x):
    """Sequential(linear, SiLU, linear): openaimodel3d.py:370-380"""
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


# Score-matrix budget (bytes) above which attention_core does not materialise softmax(q k^T) whole: the reference's
# plain path (attention.py:101-125) needs 80 x 9216 x 9216 fp32 = 27 GB per level-0 attention at the 1024 config
# (SURVEY 8a12), which is why the reference itself switches to a fused online-softmax kernel there
# (xformers.memory_efficient_attention, attention.py:146-209). Above the budget the oracle does the same with torch's
# CPU counterpart, F.scaled_dot_product_attention ("sdpa"), or walks (batch*head, query) chunks of the plain path
# ("chunked": rows of the score matrix are independent). Both are pinned to the plain path in
# tests/test_oracle_golden.py::test_large_attention_paths_match_plain.
ATTN_CHUNK_BYTES = 1 << 30
ATTN_LARGE_IMPL = "sdpa"


def attention_core(q, k, v, heads, scale, chunk_bytes=None, impl=None):
    """attention.py:101-125: per-head softmax(q k^T * scale) v on [b, n, h*d] tensors."""
    b, n, _ = q.shape
    d = q.shape[-1] // heads
    qh = q.reshape(b, n, heads, d).transpose(1, 2)
    kh = k.reshape(b, -1, heads, d).transpose(1, 2)
    vh = v.reshape(b, -1, heads, d).transpose(1, 2)
    L = kh.shape[2]
    budget = ATTN_CHUNK_BYTES if chunk_bytes is None else chunk_bytes
    if 4 * b * heads * n * L <= budget:
        sim = torch.matmul(qh, kh.transpose(-1, -2)) * scale
        p = sim.softmax(dim=-1)
        return torch.matmul(p, vh).transpose(1, 2).reshape(b, n, heads * d)
    if (ATTN_LARGE_IMPL if impl is None else impl) == "sdpa":
        o = F.scaled_dot_product_attention(qh, kh, vh, scale=scale)
        return o.transpose(1, 2).reshape(b, n, heads * d)
    # chunked: rows of the score matrix are independent, so any (batch*head, query-range) partition is exact
    qf, kf, vf = qh.reshape(b * heads, n, d), kh.reshape(b * heads, L, d), vh.reshape(b * heads, L, d)
    out = torch.empty_like(qf)
    qc = max(1, min(n, budget // (4 * L)))              # query rows per chunk for one (batch, head)
    gc = max(1, budget // (4 * L * qc))                 # (batch, head) pairs per chunk
    for g0 in range(0, b * heads, gc):
        kt = kf[g0:g0 + gc].transpose(-1, -2)
        for q0 in range(0, n, qc):
            sim = torch.matmul(qf[g0:g0 + gc, q0:q0 + qc], kt) * scale
            out[g0:g0 + gc, q0:q0 + qc] = torch.matmul(sim.softmax(dim=-1), vf[g0:g0 + gc])
    return out.reshape(b, heads, n, d).transpose(1, 2).reshape(b, n, heads * d)


def cross_attention(sd, p, x, context, heads, dim_head, cfg, image_cross):
    """CrossAttention.forward attention.py:81-144 (no mask, no relative position)."""
    scale = dim_head ** -0.5
    q = _lin(sd, p + ".to_q", x)
    if context is None:
        out = attention_core(q, _lin(sd, p + ".to_k", x), _lin(sd, p + ".to_v", x), heads, scale)
    else:
        text = context[:, :cfg.text_context_len]
        out = attention_core(q, _lin(sd, p + ".to_k", text), _lin(sd, p + ".to_v", text), heads, scale)
        if image_cross:
            img = context[:, cfg.text_context_len:]
            out_ip = attention_core(q, _lin(sd, p + ".to_k_ip", img), _lin(sd, p + ".to_v_ip", img), heads, scale)
            if cfg.image_cross_attention_scale_learnable:
                out = out + 1.0 * out_ip * (torch.tanh(sd[p + ".alpha"]) + 1)      # attention.py:139-140
            else:
                out = out + 1.0 * out_ip
    return _lin(sd, p + ".to_out.0", out)


def feed_forward(sd, p, x):
    """FeedForward with GEGLU attention.py:415-442"""
    h = _lin(sd, p + ".net.0.proj", x)
    val, gate = h.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", val * F.gelu(gate))


def transformer_block(sd, p, x, context, heads, dim_head, cfg, image_cross):
    """BasicTransformerBlock._forward attention.py:242-246 (attn1 is always self-attention here)."""
    def ln(name, t):
        return F.layer_norm(t, (t.shape[-1],), sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], 1e-5)
    x = cross_attention(sd, p + ".attn1", ln("norm1", x), None, heads, dim_head, cfg, False) + x
    x = cross_attention(sd, p + ".attn2", ln("norm2", x), context, heads, dim_head, cfg, image_cross) + x
    x = feed_forward(sd, p + ".ff", ln("norm3", x)) + x
    return x


def spatial_transformer(sd, p, x, context, heads, dim_head, cfg):
    """SpatialTransformer.forward attention.py:294-310 (use_linear=True)."""
    b, c, h, w = x.shape
    x_in = x
    x = _gn(sd, p + ".norm", x, 1e-6)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = _lin(sd, p + ".proj_in", x)
    x = transformer_block(sd, p + ".transformer_blocks.0", x, context, heads, dim_head, cfg, cfg.image_cross_attention)
    x = _lin(sd, p + ".proj_out", x)
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return x + x_in


def temporal_transformer(sd, p, x, heads, dim_head, cfg):
    """TemporalTransformer.forward attention.py:365-412 (only_self_att). x: [b, c, t, h, w].
    proj_in/out are nn.Linear when use_linear, else Conv1d(k=1) (init_attn: openaimodel3d.py:390-399)."""
    b, c, t, h, w = x.shape
    x_in = x
    x = _gn(sd, p + ".norm", x, 1e-6)
    x = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, t, c)          # (b h w) t c
    w_in = sd[p + ".proj_in.weight"]
    x = F.linear(x, w_in.reshape(w_in.shape[0], -1), sd[p + ".proj_in.bias"])
    x = transformer_block(sd, p + ".transformer_blocks.0", x, None, heads, dim_head, cfg, False)
    w_out = sd[p + ".proj_out.weight"]
    x = F.linear(x, w_out.reshape(w_out.shape[0], -1), sd[p + ".proj_out.bias"])
    x = x.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2)
    return x + x_in


def temporal_conv_block(sd, p, x):
    """TemporalConvBlock.forward openaimodel3d.py:272-279. x: [b, c, t, h, w]; GroupNorm spans (t,h,w)."""
    identity = x
    for i, conv_idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
        x = F.silu(_gn(sd, f"{p}.conv{i}.0", x, 1e-5))
        x = F.conv3d(x, sd[f"{p}.conv{i}.{conv_idx}.weight"], sd[f"{p}.conv{i}.{conv_idx}.bias"], padding=(1, 0, 0))
    return identity + x


def res_block(sd, p, x, emb, batch_size, cfg):
    """ResBlock._forward openaimodel3d.py:210-236 (no up/down, no scale-shift norm)."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5))
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    emb_out = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    h = x + h
    if cfg.temporal_conv and (p + ".temopral_conv.conv1.0.weight") in sd:
        bt, c, hh, ww = h.shape
        h5 = h.reshape(batch_size, bt // batch_size, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
    return h


def build_plan(cfg):
    """Block structure exactly as UNetModel.__init__ lays it out (openaimodel3d.py:384-546).
    Returns (input_blocks, middle, output_blocks); each block is a list of (kind, info) with the layer's index
    inside its TimestepEmbedSequential."""
    mc = cfg.model_channels
    inp = [[("conv_in", dict(cin=cfg.in_channels, cout=mc))]]
    chans = [mc]
    ch, ds = mc, 1
    nlev = len(cfg.channel_mult)
    def attn_layers(ch):
        heads = ch // cfg.num_head_channels
        ls = [("spatial", dict(ch=ch, heads=heads))]
        if cfg.temporal_attention:
            ls.append(("temporal", dict(ch=ch, heads=heads)))
        return ls
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", dict(cin=ch, cout=mult * mc, tconv=cfg.temporal_conv))]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers += attn_layers(ch)
            inp.append(layers)
            chans.append(ch)
        if level != nlev - 1:
            inp.append([("down", dict(ch=ch))])
            chans.append(ch)
            ds *= 2
    mid = [("res", dict(cin=ch, cout=ch, tconv=cfg.temporal_conv))] + attn_layers(ch) + \
          [("res", dict(cin=ch, cout=ch, tconv=cfg.temporal_conv))]
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", dict(cin=ch + ich, cout=mult * mc, tconv=cfg.temporal_conv))]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers += attn_layers(ch)
            if level and i == cfg.num_res_blocks:
                layers.append(("up", dict(ch=ch)))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def _run_block(sd, prefix, layers, h, emb, context, b, cfg):
    """TimestepEmbedSequential.forward openaimodel3d.py:36-48"""
    for j, (kind, info) in enumerate(layers):
        p = f"{prefix}.{j}"
        if kind == "conv_in":
            h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif kind == "res":
            h = res_block(sd, p, h, emb, b, cfg)
        elif kind == "spatial":
            h = spatial_transformer(sd, p, h, context, info["heads"], cfg.num_head_channels, cfg)
        elif kind == "temporal":
            bt, c, hh, ww = h.shape
            h5 = h.reshape(b, bt // b, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, p, h5, info["heads"], cfg.num_head_channels, cfg)
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        elif kind == "down":
            h = F.conv2d(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=2, padding=1)
        elif kind == "up":
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
        else:
            raise ValueError(kind)
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, timesteps, context, fs=None):
    """UNetModel.forward openaimodel3d.py:548-603. x [b, c, t, h, w] fp32; context [b, 77 + t*16, D] or [b, L, D]."""
    b, _, t, _, _ = x.shape
    mc = cfg.model_channels
    emb = _mlp(sd, "time_embed", timestep_embedding(timesteps, mc))
    l_context = context.shape[1]
    if l_context == 77 + t * 16:                                       # hard-coded in the reference :556
        ctx_text, ctx_img = context[:, :77], context[:, 77:]
        ctx_text = ctx_text.repeat_interleave(t, dim=0)
        ctx_img = ctx_img.reshape(b, t, -1, ctx_img.shape[-1]).reshape(b * t, -1, ctx_img.shape[-1])
        context = torch.cat([ctx_text, ctx_img], dim=1)
    else:
        context = context.repeat_interleave(t, dim=0)
    emb = emb.repeat_interleave(t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])
    if cfg.fs_condition:
        if fs is None:
            fs = torch.tensor([cfg.default_fs] * b, dtype=torch.long)
        fs_emb = _mlp(sd, "fps_embedding", timestep_embedding(fs, mc))
        emb = emb + fs_emb.repeat_interleave(t, dim=0)

    inp, mid, out = build_plan(cfg)
    hs = []
    for i, layers in enumerate(inp):
        h = _run_block(sd, f"input_blocks.{i}", layers, h, emb, context, b, cfg)
        if i == 0 and cfg.addition_attention:
            bt, c, hh, ww = h.shape
            h5 = h.reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, "init_attn.0", h5, 8, cfg.num_head_channels, cfg)
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        hs.append(h)
    h = _run_block(sd, "middle_block", mid, h, emb, context, b, cfg)
    for i, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f"output_blocks.{i}", layers, h, emb, context, b, cfg)
    h = F.silu(_gn(sd, "out.0", h, 1e-5))
    y = F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def unet_param_shapes(cfg):
    """name -> shape of every UNet parameter, in the reference's naming (used to synthesise weights and to
    check checkpoint-key compatibility of the host classes)."""
    shapes = {}
    mc = cfg.model_channels
    ted = mc * 4
    def lin(p, i, o, bias=True):
        shapes[p + ".weight"] = (o, i)
        if bias:
            shapes[p + ".bias"] = (o,)
    def norm(p, c):
        shapes[p + ".weight"] = (c,); shapes[p + ".bias"] = (c,)
    def conv(p, i, o, k):
        shapes[p + ".weight"] = (o, i) + k; shapes[p + ".bias"] = (o,)
    def tblock(p, dim, ctx_dim, image_cross):
        inner = dim
        for a, cd, ip in (("attn1", dim, False), ("attn2", ctx_dim, image_cross)):
            lin(f"{p}.{a}.to_q", dim, inner, False); lin(f"{p}.{a}.to_k", cd, inner, False)
            lin(f"{p}.{a}.to_v", cd, inner, False); lin(f"{p}.{a}.to_out.0", inner, dim)
            if ip:
                lin(f"{p}.{a}.to_k_ip", cd, inner, False); lin(f"{p}.{a}.to_v_ip", cd, inner, False)
                if cfg.image_cross_attention_scale_learnable:
                    shapes[f"{p}.{a}.alpha"] = ()
        lin(f"{p}.ff.net.0.proj", dim, dim * 8); lin(f"{p}.ff.net.2", dim * 4, dim)
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{p}.{n}", dim)
    def spatial(p, ch):
        norm(p + ".norm", ch); lin(p + ".proj_in", ch, ch)
        tblock(p + ".transformer_blocks.0", ch, cfg.context_dim, cfg.image_cross_attention)
        lin(p + ".proj_out", ch, ch)
    def temporal(p, ch, inner, linear):
        norm(p + ".norm", ch)
        if linear:
            lin(p + ".proj_in", ch, inner); lin(p + ".proj_out", inner, ch)
        else:
            conv(p + ".proj_in", ch, inner, (1,)); conv(p + ".proj_out", inner, ch, (1,))
        tblock(p + ".transformer_blocks.0", inner, inner, False)
    def res(p, cin, cout, tconv):
        norm(p + ".in_layers.0", cin); conv(p + ".in_layers.2", cin, cout, (3, 3))
        lin(p + ".emb_layers.1", ted, cout)
        norm(p + ".out_layers.0", cout); conv(p + ".out_layers.3", cout, cout, (3, 3))
        if cin != cout:
            conv(p + ".skip_connection", cin, cout, (1, 1))
        if tconv:
            for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
                norm(f"{p}.temopral_conv.conv{i}.0", cout); conv(f"{p}.temopral_conv.conv{i}.{ci}", cout, cout, (3, 1, 1))
    lin("time_embed.0", mc, ted); lin("time_embed.2", ted, ted)
    if cfg.fs_condition:
        lin("fps_embedding.0", mc, ted); lin("fps_embedding.2", ted, ted)
    if cfg.addition_attention:
        temporal("init_attn.0", mc, 8 * cfg.num_head_channels, False)
    inp, mid, out = build_plan(cfg)
    def block(prefix, layers):
        for j, (kind, info) in enumerate(layers):
            p = f"{prefix}.{j}"
            if kind == "conv_in":
                conv(p, info["cin"], info["cout"], (3, 3))
            elif kind == "res":
                res(p, info["cin"], info["cout"], info["tconv"])
            elif kind == "spatial":
                spatial(p, info["ch"])
            elif kind == "temporal":
                temporal(p, info["ch"], info["ch"], cfg.use_linear)
            elif kind == "down":
                conv(p + ".op", info["ch"], info["ch"], (3, 3))
            elif kind == "up":
                conv(p + ".conv", info["ch"], info["ch"], (3, 3))
    for i, layers in enumerate(inp):
        block(f"input_blocks.{i}", layers)
    block("middle_block", mid)
    for i, layers in enumerate(out):
        block(f"output_blocks.{i}", layers)
    norm("out.0", mc); conv("out.2", mc, cfg.out_channels, (3, 3))
    return shapes
