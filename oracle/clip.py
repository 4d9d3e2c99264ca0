"""CPU fp32 restatement of the frozen OpenCLIP ViT-H/14 towers and the image preprocessing in front of them
(TEST INFRASTRUCTURE - see oracle/__init__.py).

PARITY UNPINNED against the third-party code: the arithmetic lives in `open_clip_torch==2.22.0` (model `ViT-H-14`:
vision width 1280 / 32 layers / 16 heads x 80 / patch 14 / 224 px, text width 1024 / 24 layers / 16 heads / 77 tokens /
vocab 49408, pre-LN ResidualAttentionBlocks on nn.MultiheadAttention, exact GELU, LayerNorm eps 1e-5) and in `kornia`
(unpinned in the reference's requirements.txt), neither of which is in this image or under /root/reference. What is
restated here is their published algorithm, anchored on the reference's call sites:
  text    lvdm/modules/encoders/condition.py:216-234  token_embedding + positional_embedding -> resblocks[: n - layer_idx]
          with the causal attn_mask -> ln_final            (layer "penultimate": the last block is skipped)
  vision  condition.py:322-330, 345-372  preprocess (kornia resize bicubic/align_corners/antialias, (x+1)/2, mean/std) ->
          conv1 patches -> [class | patches] + positional_embedding -> ln_pre -> all resblocks   (no ln_post / proj)
The attention block is pinned against torch.nn.MultiheadAttention itself (tests/test_oracle_golden.py), the resize
against torch's own bicubic interpolate + conv2d; the composition order is the reference's, read from the lines above.
"""
import math

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def tower_shapes(prefix, width, layers, mlp_ratio=4.0):
    t = {}
    hid = int(width * mlp_ratio)
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}"
        t[p + ".ln_1.weight"] = (width,); t[p + ".ln_1.bias"] = (width,)
        t[p + ".attn.in_proj_weight"] = (3 * width, width); t[p + ".attn.in_proj_bias"] = (3 * width,)
        t[p + ".attn.out_proj.weight"] = (width, width); t[p + ".attn.out_proj.bias"] = (width,)
        t[p + ".ln_2.weight"] = (width,); t[p + ".ln_2.bias"] = (width,)
        t[p + ".mlp.c_fc.weight"] = (hid, width); t[p + ".mlp.c_fc.bias"] = (hid,)
        t[p + ".mlp.c_proj.weight"] = (width, hid); t[p + ".mlp.c_proj.bias"] = (width,)
    return t


def clip_text_shapes(width=1024, layers=24, context=77, vocab=49408, embed_dim=1024):
    """state_dict of the CLIP object after `del model.visual` (FrozenOpenCLIPEmbedder, condition.py:188-190), keys
    relative to `cond_stage_model.`"""
    t = {"model.positional_embedding": (context, width), "model.text_projection": (width, embed_dim),
         "model.logit_scale": (), "model.token_embedding.weight": (vocab, width),
         "model.ln_final.weight": (width,), "model.ln_final.bias": (width,)}
    t.update(tower_shapes("model.transformer", width, layers))
    return t


def clip_vision_shapes(width=1280, layers=32, patch=14, image=224, embed_dim=1024, text_width=1024, context=77, vocab=49408):
    """state_dict of the CLIP object after `del model.transformer` (FrozenOpenCLIPImageEmbedderV2, condition.py:303-306):
    the vision tower plus the text-side leftovers the CLIP class still owns."""
    n = (image // patch) ** 2 + 1
    t = {"model.positional_embedding": (context, text_width), "model.text_projection": (text_width, embed_dim),
         "model.logit_scale": (), "model.visual.class_embedding": (width,), "model.visual.positional_embedding": (n, width),
         "model.visual.proj": (width, embed_dim), "model.visual.conv1.weight": (width, 3, patch, patch),
         "model.visual.ln_pre.weight": (width,), "model.visual.ln_pre.bias": (width,),
         "model.visual.ln_post.weight": (width,), "model.visual.ln_post.bias": (width,),
         "model.token_embedding.weight": (vocab, text_width),
         "model.ln_final.weight": (text_width,), "model.ln_final.bias": (text_width,)}
    t.update(tower_shapes("model.visual.transformer", width, layers))
    return t


def mha(sd, p, x, heads, mask=None):
    """nn.MultiheadAttention forward (batch-first restatement): q scaled by d^-1/2, additive mask, softmax, out_proj."""
    B, L, D = x.shape
    d = D // heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    sp = lambda t: t.reshape(B, L, heads, d).transpose(1, 2)
    q, k, v = sp(q) * d ** -0.5, sp(k), sp(v)
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    o = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def resblock(sd, p, x, heads, mask=None):
    """open_clip ResidualAttentionBlock (no layer scale): x + attn(ln_1 x); x + c_proj(gelu(c_fc(ln_2 x)))"""
    ln = lambda n, t: F.layer_norm(t, (t.shape[-1],), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)
    x = x + mha(sd, p + ".attn", ln("ln_1", x), heads, mask)
    h = F.gelu(F.linear(ln("ln_2", x), sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"]))
    return x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])


def causal_mask(n):
    """open_clip build_attention_mask: -inf strictly above the diagonal"""
    return torch.full((n, n), float("-inf")).triu_(1)


@torch.no_grad()
def text_forward(sd, tokens, heads=16, layer_idx=1):
    """FrozenOpenCLIPEmbedder.encode_with_transformer condition.py:215-222; tokens int64 [B, 77]."""
    x = sd["model.token_embedding.weight"][tokens] + sd["model.positional_embedding"]
    n = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("model.transformer.resblocks."))
    mask = causal_mask(tokens.shape[1])
    for i in range(n - layer_idx):                                     # text_transformer_forward :224-231
        x = resblock(sd, f"model.transformer.resblocks.{i}", x, heads, mask)
    return F.layer_norm(x, (x.shape[-1],), sd["model.ln_final.weight"], sd["model.ln_final.bias"], 1e-5)


def gaussian_kernel1d(ks, sigma):
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    g = torch.exp(-x.pow(2) / (2 * sigma ** 2))
    return g / g.sum()


@torch.no_grad()
def preprocess(img, size=(224, 224), antialias=True):
    """FrozenOpenCLIPImageEmbedderV2.preprocess condition.py:322-330 with kornia.geometry.resize restated: when
    downscaling with antialias, gaussian_blur2d (separable, reflect) with sigma = max((factor-1)/2, 0.001) and
    ks = int(max(4 sigma, 3)) made odd; then F.interpolate(bicubic, align_corners=True)."""
    x = img.float()
    H, W = x.shape[-2:]
    fy, fx = H / size[0], W / size[1]
    if antialias and max(fy, fx) > 1:
        sy, sx = max((fy - 1) / 2, 0.001), max((fx - 1) / 2, 0.001)
        ky, kx = int(max(4 * sy, 3)), int(max(4 * sx, 3))
        ky += (ky % 2 == 0); kx += (kx % 2 == 0)
        C = x.shape[1]
        gx = gaussian_kernel1d(kx, sx).reshape(1, 1, 1, kx).repeat(C, 1, 1, 1)
        gy = gaussian_kernel1d(ky, sy).reshape(1, 1, ky, 1).repeat(C, 1, 1, 1)
        x = F.conv2d(F.pad(x, (kx // 2, kx // 2, 0, 0), mode="reflect"), gx, groups=C)
        x = F.conv2d(F.pad(x, (0, 0, ky // 2, ky // 2), mode="reflect"), gy, groups=C)
    x = F.interpolate(x, size=size, mode="bicubic", align_corners=True)
    x = (x + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN).reshape(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).reshape(1, 3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def vision_forward(sd, img, heads=16, size=(224, 224), antialias=True):
    """FrozenOpenCLIPImageEmbedderV2.encode_with_vision_transformer condition.py:345-372 -> [B, 1 + grid^2, width]."""
    x = preprocess(img, size, antialias)
    w = sd["model.visual.conv1.weight"]
    x = F.conv2d(x, w, stride=w.shape[-1])                              # :349 (no bias)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd["model.visual.class_embedding"].reshape(1, 1, -1).expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1) + sd["model.visual.positional_embedding"]
    x = F.layer_norm(x, (x.shape[-1],), sd["model.visual.ln_pre.weight"], sd["model.visual.ln_pre.bias"], 1e-5)
    n = 1 + max(int(k.split(".")[4]) for k in sd if k.startswith("model.visual.transformer.resblocks."))
    for i in range(n):
        x = resblock(sd, f"model.visual.transformer.resblocks.{i}", x, heads)
    return x
