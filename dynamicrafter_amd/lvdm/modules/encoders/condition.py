"""Frozen OpenCLIP ViT-H/14 text and vision towers on the HIP path.

Drop-ins for `FrozenOpenCLIPEmbedder` (lvdm/modules/encoders/condition.py:174-234) and
`FrozenOpenCLIPImageEmbedderV2` (:295-372): same constructor keywords (the YAMLs pass `freeze`, `layer`), same
`forward` / `encode` contract (text -> [B, 77, 1024] after ln_final with the LAST block skipped for
layer="penultimate"; image [B,3,H,W] in [-1,1] -> all 257 tokens [B, 257, 1280] of the full tower, no ln_post / proj),
and the same state_dict keys as the `open_clip` CLIP object the reference keeps (`model.visual.*` deleted in the text
wrapper, `model.transformer.*` deleted in the vision wrapper; text-side leftovers - token_embedding, ln_final,
text_projection, logit_scale - kept because the reference keeps them), so `cond_stage_model.*` / `embedder.*` of a
released checkpoint load with strict=True.

The reference builds the towers with open_clip.create_model_and_transforms(arch, pretrained=version) (:188, :303); that
fetches third-party weights and is impossible offline. Here the architecture (open_clip_torch 2.22.0 `ViT-H-14`) is
built from its published hyper-parameters with random initial weights; real weights arrive through the checkpoint.
Every dense layer is dc_gemm_conv (bias / GELU / residual epilogues) or dc_layernorm; attention is dc_attn_small
(16 heads x 80 in the vision tower, 16 x 64 with the causal mask in the text tower); the image preprocessing
(kornia bicubic-224 with antialias, CLIP mean/std) is dc_clip_preprocess; dc_patchify / dc_embed_tokens feed the
first GEMM. Once per clip: <0.1 % of a clip's FLOPs.

Tokenisation needs CLIP's BPE vocabulary file (`bpe_simple_vocab_16e6.txt.gz`, shipped inside the open_clip / CLIP
packages, not in this image): pass `bpe_path=` or set DC_CLIP_BPE; without it `forward(list of str)` raises and
`encode_with_transformer(token ids)` is the entry point.
"""
import gzip
import html
import os

import torch
import torch.nn as nn

from .... import ops
from ....ops import PackedWeight
from ....param_tree import attach_params

_BF16 = torch.bfloat16

# open_clip_torch 2.22.0 model_configs/ViT-H-14.json
ARCHS = {
    "ViT-H-14": dict(embed_dim=1024,
                     vision=dict(image_size=224, layers=32, width=1280, heads=16, patch_size=14, mlp_ratio=4.0),
                     text=dict(context_length=77, vocab_size=49408, width=1024, heads=16, layers=24, mlp_ratio=4.0)),
}


def _arch(arch):
    if isinstance(arch, dict):
        return arch
    if arch not in ARCHS:
        raise NotImplementedError(f"OpenCLIP arch {arch!r}: only {sorted(ARCHS)} is built (the released configs use ViT-H-14)")
    return ARCHS[arch]


def _tower_shapes(prefix, width, layers, mlp_ratio):
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


class _Tower(nn.Module):
    """Shared plumbing: parameter lookup, weight repacking, the pre-LN transformer block sequence on bf16 rows."""

    def _p(self, name):
        node = self
        parts = name.split(".")
        for s in parts[:-1]:
            node = node._modules[s]
        return node._parameters[parts[-1]]

    def _buf(self, tag, rows, cols, dtype=_BF16, device=None):
        return self._arena.get(tag, rows, cols, dtype, device)

    def _pack_blocks(self, prefix, layers, device):
        f32 = lambda n: self._p(n).detach().to(device=device, dtype=torch.float32).contiguous()
        lin = lambda w, b: PackedWeight.linear(self._p(w), self._p(b), device)
        out = []
        for i in range(layers):
            p = f"{prefix}.resblocks.{i}"
            out.append({"ln1": (f32(p + ".ln_1.weight"), f32(p + ".ln_1.bias")),
                        "qkv": lin(p + ".attn.in_proj_weight", p + ".attn.in_proj_bias"),
                        "o": lin(p + ".attn.out_proj.weight", p + ".attn.out_proj.bias"),
                        "ln2": (f32(p + ".ln_2.weight"), f32(p + ".ln_2.bias")),
                        "fc": lin(p + ".mlp.c_fc.weight", p + ".mlp.c_fc.bias"),
                        "proj": lin(p + ".mlp.c_proj.weight", p + ".mlp.c_proj.bias")})
        return out

    def _blocks(self, x, blocks, *, B, L, heads, causal):
        """open_clip ResidualAttentionBlock x n: x += out_proj(attn(in_proj(ln_1 x))); x += c_proj(gelu(c_fc(ln_2 x)))"""
        M, D = x.shape
        d = D // heads
        dev = x.device
        for W in blocks:
            n = ops.layernorm(x, self._buf("ln", M, D, device=dev), *W["ln1"])
            qkv = ops.gemm(n, W["qkv"], self._buf("qkv", M, 3 * D, device=dev))
            att = self._buf("att", M, D, device=dev)
            ops.attn_small(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], att, batch=B, heads=heads, Lq=L, Lk=L, d=d,
                           scale=float(d) ** -0.5, causal=causal)
            ops.gemm(att, W["o"], x, residual=x)
            n = ops.layernorm(x, self._buf("ln", M, D, device=dev), *W["ln2"])
            h = ops.gemm(n, W["fc"], self._buf("mlp", M, W["fc"].N, device=dev), gelu=True)
            ops.gemm(h, W["proj"], x, residual=x)
        return x

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False


class FrozenOpenCLIPEmbedder(_Tower):
    """Text tower. `forward(list of str | int64 token ids [B, 77]) -> fp32 [B, 77, width]`."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 layer="last", bpe_path=None):
        super().__init__()
        assert layer in self.LAYERS
        a = _arch(arch)
        t = a["text"]
        self.width, self.heads, self.n_layers = t["width"], t["heads"], t["layers"]
        self.context, self.vocab = t["context_length"], t["vocab_size"]
        if self.width % 64 or (self.width // self.heads) % 2:
            raise NotImplementedError("text width must be a multiple of 64")
        shapes = {"model.positional_embedding": (self.context, self.width),
                  "model.text_projection": (self.width, a["embed_dim"]), "model.logit_scale": (),
                  "model.token_embedding.weight": (self.vocab, self.width),
                  "model.ln_final.weight": (self.width,), "model.ln_final.bias": (self.width,)}
        shapes.update(_tower_shapes("model.transformer", self.width, self.n_layers, t["mlp_ratio"]))
        attach_params(self, shapes)
        self.device = device
        self.max_length = max_length
        self.layer = layer
        self.layer_idx = 0 if layer == "last" else 1
        self._bpe_path = bpe_path
        self._tokenizer = None
        self._packed = None
        self._arena = ops.Arena()
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_packed", None))
        if freeze:
            self.freeze()

    def packed(self, device):
        if self._packed is None or self._packed["device"] != device:
            bf = lambda n: self._p(n).detach().to(device=device, dtype=_BF16).contiguous()
            f32 = lambda n: self._p(n).detach().to(device=device, dtype=torch.float32).contiguous()
            self._packed = {"device": device, "table": bf("model.token_embedding.weight"),
                            "pos": bf("model.positional_embedding"),
                            "ln_final": (f32("model.ln_final.weight"), f32("model.ln_final.bias")),
                            "blocks": self._pack_blocks("model.transformer", self.n_layers, device)}
        return self._packed

    def tokenize(self, text):
        if self._tokenizer is None:
            self._tokenizer = SimpleTokenizer(self._bpe_path or os.environ.get("DC_CLIP_BPE"))
        return self._tokenizer(text, self.context)

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        """condition.py:215-222 (+ :224-231): int64 token ids [B, context] on the GPU."""
        if not tokens.is_cuda:
            raise RuntimeError("FrozenOpenCLIPEmbedder runs on the HIP path only (no CPU fallback)")
        dev = tokens.device
        P = self.packed(dev)
        B, L = tokens.shape
        if L != self.context:
            raise ValueError(f"expected {self.context} tokens per prompt, got {L}")
        x = self._buf("x", B * L, self.width, device=dev)
        ops.embed_tokens(tokens.to(torch.int64).contiguous(), P["table"], P["pos"], x)
        x = self._blocks(x, P["blocks"][:self.n_layers - self.layer_idx], B=B, L=L, heads=self.heads, causal=True)
        y = ops.layernorm(x, self._buf("y", B * L, self.width, device=dev), *P["ln_final"])
        out = torch.empty((B, L, self.width), dtype=torch.float32, device=dev)
        ops.rows_to_nchw(y, out, N=B * L, Cc=self.width, HW=1)
        return out

    def forward(self, text):
        if isinstance(text, torch.Tensor):
            return self.encode_with_transformer(text)
        dev = self._p("model.ln_final.weight").device
        return self.encode_with_transformer(self.tokenize(text).to(dev))

    def encode(self, text):
        return self(text)


class FrozenOpenCLIPImageEmbedderV2(_Tower):
    """Vision tower. `forward(image [B, 3, H, W] in [-1, 1]) -> fp32 [B, 1 + (224/14)^2, width]`."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled",
                 antialias=True):
        super().__init__()
        if layer == "penultimate":
            raise NotImplementedError()                                 # as the reference (condition.py:312-314)
        a = _arch(arch)
        v, t = a["vision"], a["text"]
        self.width, self.heads, self.n_layers = v["width"], v["heads"], v["layers"]
        self.patch, self.image_size = v["patch_size"], v["image_size"]
        self.grid = self.image_size // self.patch
        self.n_tokens = self.grid * self.grid + 1
        if self.width % 64 or (self.width // self.heads) % 2:
            raise NotImplementedError("vision width must be a multiple of 64")
        shapes = {"model.positional_embedding": (t["context_length"], t["width"]),
                  "model.text_projection": (t["width"], a["embed_dim"]), "model.logit_scale": (),
                  "model.visual.class_embedding": (self.width,),
                  "model.visual.positional_embedding": (self.n_tokens, self.width),
                  "model.visual.proj": (self.width, a["embed_dim"]),
                  "model.visual.conv1.weight": (self.width, 3, self.patch, self.patch),
                  "model.visual.ln_pre.weight": (self.width,), "model.visual.ln_pre.bias": (self.width,)}
        shapes.update(_tower_shapes("model.visual.transformer", self.width, self.n_layers, v["mlp_ratio"]))
        shapes.update({"model.visual.ln_post.weight": (self.width,), "model.visual.ln_post.bias": (self.width,),
                       "model.token_embedding.weight": (t["vocab_size"], t["width"]),
                       "model.ln_final.weight": (t["width"],), "model.ln_final.bias": (t["width"],)})
        attach_params(self, shapes)
        self.device = device
        self.layer = layer
        self.antialias = antialias
        self.register_buffer("mean", torch.Tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer("std", torch.Tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)
        self._packed = None
        self._arena = ops.Arena()
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_packed", None))
        if freeze:
            self.freeze()

    def packed(self, device):
        if self._packed is None or self._packed["device"] != device:
            f32 = lambda n: self._p(n).detach().to(device=device, dtype=torch.float32).contiguous()
            pos = self._p("model.visual.positional_embedding").detach().float()
            cls = self._p("model.visual.class_embedding").detach().float()
            self._packed = {"device": device,
                            "conv1": PackedWeight.linear(self._p("model.visual.conv1.weight"), None, device),
                            "cls_pos0": (cls + pos[0]).reshape(1, -1).to(device=device, dtype=_BF16).contiguous(),
                            "pos": pos[1:].to(device=device, dtype=_BF16).contiguous(),
                            "ln_pre": (f32("model.visual.ln_pre.weight"), f32("model.visual.ln_pre.bias")),
                            "blocks": self._pack_blocks("model.visual.transformer", self.n_layers, device)}
        return self._packed

    def preprocess(self, x):
        """condition.py:322-330"""
        return ops.clip_preprocess(x, (self.image_size, self.image_size), self.antialias, tuple(self.mean.tolist()),
                                   tuple(self.std.tolist()))

    @torch.no_grad()
    def encode_with_vision_transformer(self, x):
        """condition.py:345-372"""
        if not x.is_cuda:
            raise RuntimeError("FrozenOpenCLIPImageEmbedderV2 runs on the HIP path only (no CPU fallback)")
        dev = x.device
        P = self.packed(dev)
        x = self.preprocess(x)
        B = x.shape[0]
        g2, L, D = self.grid * self.grid, self.n_tokens, self.width
        patches = self._buf("patches", B * g2, P["conv1"].K, device=dev)
        ops.patchify(x, patches, patch=self.patch)
        tok = self._buf("tok", B * L, D, device=dev)
        for b in range(B):
            body = tok[b * L + 1:(b + 1) * L]
            ops.gemm(patches[b * g2:(b + 1) * g2], P["conv1"], body)         # conv1 (no bias) as a GEMM over patches
            ops.add_rows(body, P["pos"], body)                                # + positional_embedding[1:]
            ops.copy2d(P["cls_pos0"], tok[b * L:b * L + 1])                   # class_embedding + positional_embedding[0]
        h = ops.layernorm(tok, self._buf("x", B * L, D, device=dev), *P["ln_pre"])
        h = self._blocks(h, P["blocks"], B=B, L=L, heads=self.heads, causal=False)
        out = torch.empty((B, L, D), dtype=torch.float32, device=dev)
        ops.rows_to_nchw(h, out, N=B * L, Cc=D, HW=1)
        return out

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    def encode(self, image):
        return self(image)


# ------------------------------------------------------------------------------------------------------------------
class SimpleTokenizer:
    """CLIP's byte-level BPE (lower-cased, `</w>` word ends, <start_of_text> ... <end_of_text>, zero padded to the
    context length) - the algorithm open_clip.tokenize applies (condition.py:211). Needs the merges file."""

    def __init__(self, bpe_path):
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError(
                "CLIP BPE vocabulary not found: pass bpe_path= / set DC_CLIP_BPE to bpe_simple_vocab_16e6.txt.gz (it ships "
                "inside the open_clip / CLIP python packages), or call encode_with_transformer() with token ids")
        import regex
        # printable stand-ins for the 256 byte values
        keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
        chars, extra = keep[:], 0
        for b in range(256):
            if b not in keep:
                keep.append(b); chars.append(256 + extra); extra += 1
        self.byte_enc = {b: chr(c) for b, c in zip(keep, chars)}
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in lines[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_enc.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.enc = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.cache = {}
        self.pat = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                 regex.IGNORECASE)
        self.sot, self.eot = self.enc["<start_of_text>"], self.enc["<end_of_text>"]

    def _bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            pairs = {(a, b) for a, b in zip(word, word[1:])}
            best = min(pairs, key=lambda p: self.rank.get(p, float("inf")))
            if best not in self.rank:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1]); i += 2
                else:
                    merged.append(word[i]); i += 1
            word = merged
        self.cache[token] = word
        return word

    def encode(self, text):
        text = " ".join(html.unescape(html.unescape(text)).split()).strip().lower()
        ids = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_enc[b] for b in tok.encode("utf-8"))
            ids.extend(self.enc[t] for t in self._bpe(tok))
        return ids

    def __call__(self, texts, context_length=77):
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids)
        return out
