"""UNetModel — the 3D denoiser of DynamiCrafter, executed as a sequence of HIP kernels on MI355X.

Drop-in for the reference class of the same dotted path (lvdm/modules/networks/openaimodel3d.py:281-603): same
constructor keywords (configs/*.yaml `unet_config.params`), same `forward(x, timesteps, context, features_adapter,
fs, **kwargs)` contract ([B,8,T,h,w] -> [B,4,T,h,w]), same state_dict keys. Inside, nothing of the reference's
module graph exists: activations live as channels-last bf16 rows [(b t h w), C] in HBM, every layer is one or a
few calls into libdcrafter_hip.so (include/dcrafter_hip.h), and all scratch is pre-allocated so that a whole
forward is hipGraph-capturable.

Layer recipes (reference line numbers in openaimodel3d.py / attention.py):
  ResBlock :210-236           GN+SiLU -> conv3x3(+bias +emb[b]) -> GN+SiLU -> conv3x3(+bias +skip) -> TemporalConvBlock
  TemporalConvBlock :272-279  4 x [GN over (t,h,w)+SiLU -> 3-tap temporal conv], + identity (fused in the last conv)
  SpatialTransformer attention.py:294-310; BasicTransformerBlock :242-246; CrossAttention :81-144; GEGLU :415-422
  TemporalTransformer attention.py:365-412 (both attentions are self-attention over the T frames)
"""
import math

import torch
import torch.nn as nn

from .... import ops
from ....ops import PackedWeight
from ....param_tree import attach_params

import os

_BF16 = torch.bfloat16
_FF_FUSED = os.environ.get("DC_FF_FUSED", "1") != "0"
_LN_FUSED = os.environ.get("DC_LN_FUSED", "1") != "0"      # A/B switches of the fused kernels in ff_fused.hip
_LR_FUSED = os.environ.get("DC_LR_FUSED", "1") != "0"      # Linear + residual at K <= 640 in the X-stationary kernel
_TA_FUSED = os.environ.get("DC_TA_FUSED", "1") != "0"
_TA_FUSED_C = tuple(int(k) for k in os.environ.get("DC_TA_FUSED_C", "320,640").split(","))      # widths of the fused temporal attention
_FFP_FUSED = os.environ.get("DC_FFP_FUSED", "1") != "0"
_TC_FUSED = os.environ.get("DC_TC_FUSED", "1") != "0"
# below this row count the tile GEMMs + norm kernels are used (the 1024 config's level-0/1 tensors have >= 73728 rows; at the
# 512 / 256 configs level 1 has 20480 / 8192: fused kernels measured 42.1 -> 41.8 and 24.1 -> 23.4 ms per step there)
_FUSED_MIN_ROWS = int(os.environ.get("DC_FUSED_MIN_ROWS", "4096"))
_LN_FUSED_K = tuple(int(k) for k in os.environ.get("DC_LN_FUSED_K", "320,640").split(","))
C_IN_PAD = 64   # conv_in consumes the 8 latent+concat channels zero-padded to one 64-wide K slice


def _block_layout(cfg):
    """(input_blocks, middle_block, output_blocks): per block a list of (kind, attrs), mirroring the construction
    order of the reference so that the numeric key segments of its state_dict are reproduced."""
    mc, mults, nrb = cfg["model_channels"], list(cfg["channel_mult"]), cfg["num_res_blocks"]
    attn_res = set(cfg["attention_resolutions"])
    hc = cfg["num_head_channels"]
    tconv = bool(cfg["temporal_conv"])

    def transformers(ch):
        out = [("spatial", {"ch": ch, "heads": ch // hc})]
        if cfg["temporal_attention"]:
            out.append(("temporal", {"ch": ch, "inner": ch, "heads": ch // hc, "linear": bool(cfg["use_linear"])}))
        return out

    down_path = [[("conv_in", {"cin": cfg["in_channels"], "cout": mc})]]
    widths = [mc]
    ch, ds = mc, 1
    for lvl, m in enumerate(mults):
        for _ in range(nrb):
            blk = [("res", {"cin": ch, "cout": m * mc, "tconv": tconv})]
            ch = m * mc
            if ds in attn_res:
                blk += transformers(ch)
            down_path.append(blk)
            widths.append(ch)
        if lvl + 1 < len(mults):
            down_path.append([("down", {"ch": ch})])
            widths.append(ch)
            ds *= 2
    middle = [("res", {"cin": ch, "cout": ch, "tconv": tconv})] + transformers(ch) + \
             [("res", {"cin": ch, "cout": ch, "tconv": tconv})]
    up_path = []
    for lvl in reversed(range(len(mults))):
        for i in range(nrb + 1):
            skip = widths.pop()
            blk = [("res", {"cin": ch + skip, "cout": mults[lvl] * mc, "tconv": tconv})]
            ch = mults[lvl] * mc
            if ds in attn_res:
                blk += transformers(ch)
            if lvl and i == nrb:
                blk.append(("up", {"ch": ch}))
                ds //= 2
            up_path.append(blk)
    return down_path, middle, up_path


def _shape_table(cfg):
    """{state_dict key: shape} in the reference's naming."""
    t = {}
    mc = cfg["model_channels"]
    emb_dim = 4 * mc

    def affine(p, c):
        t[p + ".weight"] = (c,)
        t[p + ".bias"] = (c,)

    def dense(p, i, o, bias=True):
        t[p + ".weight"] = (o, i)
        if bias:
            t[p + ".bias"] = (o,)

    def conv(p, i, o, *k):
        t[p + ".weight"] = (o, i) + tuple(k)
        t[p + ".bias"] = (o,)

    def attn(p, dim, ctx, with_ip):
        dense(p + ".to_q", dim, dim, False)
        dense(p + ".to_k", ctx, dim, False)
        dense(p + ".to_v", ctx, dim, False)
        dense(p + ".to_out.0", dim, dim)
        if with_ip:
            dense(p + ".to_k_ip", ctx, dim, False)
            dense(p + ".to_v_ip", ctx, dim, False)
            if cfg["image_cross_attention_scale_learnable"]:
                t[p + ".alpha"] = ()

    def tblock(p, dim, ctx, with_ip):
        attn(p + ".attn1", dim, dim, False)
        dense(p + ".ff.net.0.proj", dim, 8 * dim)
        dense(p + ".ff.net.2", 4 * dim, dim)
        attn(p + ".attn2", dim, ctx, with_ip)
        for n in ("norm1", "norm2", "norm3"):
            affine(f"{p}.{n}", dim)

    def temporal(p, ch, inner, linear):
        affine(p + ".norm", ch)
        if linear:
            dense(p + ".proj_in", ch, inner)
        else:
            conv(p + ".proj_in", ch, inner, 1)
        tblock(p + ".transformer_blocks.0", inner, inner, False)
        if linear:
            dense(p + ".proj_out", inner, ch)
        else:
            conv(p + ".proj_out", inner, ch, 1)

    dense("time_embed.0", mc, emb_dim)
    dense("time_embed.2", emb_dim, emb_dim)
    if cfg["fs_condition"]:
        dense("fps_embedding.0", mc, emb_dim)
        dense("fps_embedding.2", emb_dim, emb_dim)

    def emit(prefix, blk):
        for j, (kind, a) in enumerate(blk):
            p = f"{prefix}.{j}"
            if kind == "conv_in":
                conv(p, a["cin"], a["cout"], 3, 3)
            elif kind == "res":
                affine(p + ".in_layers.0", a["cin"])
                conv(p + ".in_layers.2", a["cin"], a["cout"], 3, 3)
                dense(p + ".emb_layers.1", emb_dim, a["cout"])
                affine(p + ".out_layers.0", a["cout"])
                conv(p + ".out_layers.3", a["cout"], a["cout"], 3, 3)
                if a["cin"] != a["cout"]:
                    conv(p + ".skip_connection", a["cin"], a["cout"], 1, 1)
                if a["tconv"]:
                    for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):   # Sequential index of the Conv3d (Dropout at 2)
                        affine(f"{p}.temopral_conv.conv{i}.0", a["cout"])
                        conv(f"{p}.temopral_conv.conv{i}.{ci}", a["cout"], a["cout"], 3, 1, 1)
            elif kind == "spatial":
                affine(p + ".norm", a["ch"])
                dense(p + ".proj_in", a["ch"], a["ch"])
                tblock(p + ".transformer_blocks.0", a["ch"], cfg["context_dim"], bool(cfg["image_cross_attention"]))
                dense(p + ".proj_out", a["ch"], a["ch"])
            elif kind == "temporal":
                temporal(p, a["ch"], a["inner"], a["linear"])
            elif kind == "down":
                conv(p + ".op", a["ch"], a["ch"], 3, 3)
            elif kind == "up":
                conv(p + ".conv", a["ch"], a["ch"], 3, 3)

    down_path, middle, up_path = _block_layout(cfg)
    emit("input_blocks.0", down_path[0])
    if cfg["addition_attention"]:
        temporal("init_attn.0", mc, 8 * cfg["num_head_channels"], False)
    for i, blk in enumerate(down_path[1:], start=1):
        emit(f"input_blocks.{i}", blk)
    emit("middle_block", middle)
    for i, blk in enumerate(up_path):
        emit(f"output_blocks.{i}", blk)
    affine("out.0", mc)
    conv("out.2", mc, cfg["out_channels"], 3, 3)
    return t


_Arena = ops.Arena


class UNetModel(nn.Module):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None,
                 use_scale_shift_norm=False, resblock_updown=False, num_heads=-1, num_head_channels=-1,
                 transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False,
                 tempspatial_aware=False, temporal_attention=True, use_relative_position=True,
                 use_causal_attention=False, temporal_length=None, use_fp16=False, addition_attention=False,
                 temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False):
        super().__init__()
        unsupported = []
        if dims != 2: unsupported.append("dims != 2")
        if use_scale_shift_norm: unsupported.append("use_scale_shift_norm")
        if resblock_updown: unsupported.append("resblock_updown")
        if not conv_resample: unsupported.append("conv_resample=False")
        if num_head_channels != 64: unsupported.append("num_head_channels != 64 (kernels are head_dim 64)")
        if transformer_depth != 1: unsupported.append("transformer_depth != 1")
        if not use_linear: unsupported.append("use_linear=False")
        if tempspatial_aware: unsupported.append("tempspatial_aware")
        if use_relative_position: unsupported.append("use_relative_position")
        if use_causal_attention: unsupported.append("use_causal_attention")
        if not temporal_selfatt_only: unsupported.append("temporal cross-attention")
        if context_dim is None: unsupported.append("context_dim=None")
        if temporal_length is not None and temporal_length > 16: unsupported.append("temporal_length > 16")
        if model_channels % 64 != 0: unsupported.append("model_channels % 64 != 0")
        if unsupported:
            raise NotImplementedError("UNetModel (HIP path) covers the released DynamiCrafter configurations; "
                                      "unsupported here: " + ", ".join(unsupported))
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = list(attention_resolutions)
        self.channel_mult = list(channel_mult)
        self.dropout, self.use_checkpoint = dropout, use_checkpoint        # accepted, irrelevant at inference
        self.temporal_attention, self.temporal_length = temporal_attention, temporal_length
        self.addition_attention = addition_attention
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale_learnable = image_cross_attention_scale_learnable
        self.default_fs, self.fs_condition = default_fs, fs_condition
        self.context_dim = context_dim
        self.dtype = torch.float32
        self._cfg = dict(in_channels=in_channels, out_channels=out_channels, model_channels=model_channels,
                         attention_resolutions=list(attention_resolutions), num_res_blocks=num_res_blocks,
                         channel_mult=list(channel_mult), num_head_channels=num_head_channels,
                         context_dim=context_dim, temporal_conv=temporal_conv, temporal_attention=temporal_attention,
                         addition_attention=addition_attention, image_cross_attention=image_cross_attention,
                         image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                         fs_condition=fs_condition, use_linear=use_linear)
        self._layout = _block_layout(self._cfg)
        # skip-connection plan: up block u consumes cat([h, skip]) where skip is the output of down block n-1-u. Both
        # producers write straight into their column range of one buffer per up block (no torch.cat copy,
        # reference openaimodel3d.py:596): _cat_plan[u] = (channels of h, channels of the skip)
        self._cat_plan = self._make_cat_plan()
        attach_params(self, _shape_table(self._cfg))
        self._packed = None
        self._arena = _Arena()
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_packed", None))

    def _make_cat_plan(self):
        down_path, middle, up_path = self._layout
        out_ch = lambda blk: [a.get("cout", a.get("ch")) for k, a in blk if k in ("conv_in", "res", "down", "up")][-1]
        widths = [out_ch(b) for b in down_path]
        ch = out_ch(middle)
        plan = []
        for blk in up_path:
            skip = widths.pop()
            assert blk[0][1]["cin"] == ch + skip
            plan.append((ch, skip))
            ch = out_ch(blk)
        return plan

    # ------------------------------------------------------------------ weights -> device layout
    def _p(self, name):
        node = self
        parts = name.split(".")
        for s in parts[:-1]:
            node = node._modules[s]
        return node._parameters[parts[-1]]

    def _has(self, name):
        node = self
        parts = name.split(".")
        for s in parts[:-1]:
            node = node._modules.get(s)
            if node is None:
                return False
        return parts[-1] in node._parameters

    def invalidate_packed(self):
        self._packed = None

    def _pack(self, device):
        """Derive the bf16 / fp32 device copies the kernels read (once per weight load)."""
        P = {}
        f32 = lambda n: self._p(n).detach().to(device=device, dtype=torch.float32).contiguous()
        lin = lambda n: PackedWeight.linear(self._p(n + ".weight"), self._p(n + ".bias") if self._has(n + ".bias") else None, device)

        def attn_self(p):
            w = torch.cat([self._p(f"{p}.to_{x}.weight") for x in "qkv"], dim=0)
            return {"qkv": PackedWeight.linear(w, None, device), "out": lin(p + ".to_out.0")}

        def tblock(p, cross):
            d = {"attn1": attn_self(p + ".attn1"), "ff1": lin(p + ".ff.net.0.proj"), "ff2": lin(p + ".ff.net.2")}
            if d["ff2"].N == 320 and d["ff1"].N == 2560:          # dim 320: the fused FeedForward kernel's weight order
                d["ff2p"] = ops.ff2_permuted(self._p(p + ".ff.net.2.weight"), device)
            for n in ("norm1", "norm2", "norm3"):
                d[n] = (f32(f"{p}.{n}.weight"), f32(f"{p}.{n}.bias"))
            if cross:
                a = p + ".attn2"
                d["q2"] = lin(a + ".to_q")
                names = ["to_k", "to_v"] + (["to_k_ip", "to_v_ip"] if self.image_cross_attention else [])
                d["kv_ctx"] = PackedWeight.linear(torch.cat([self._p(f"{a}.{x}.weight") for x in names], 0), None, device)
                d["out2"] = lin(a + ".to_out.0")
                d["ip_scale"] = 1.0
                if self.image_cross_attention and self.image_cross_attention_scale_learnable:
                    d["ip_scale"] = float(torch.tanh(self._p(a + ".alpha").detach().float()).item() + 1.0)
            else:
                d["attn2"] = attn_self(p + ".attn2")
            d["key"] = p                       # stable name of the block (keys the hoisted context K/V of a sampler run)
            return d

        def with_projp(d, p):
            # dim 320: proj_out in the k order of the fused FeedForward + proj_out kernel (ops.ff_geglu_proj_fused320)
            w = self._p(p + ".proj_out.weight")
            if "ff2p" in d["blk"] and w.shape[0] == 320 and w[0].numel() == 320:
                d["proj_p"] = ops.ff2_permuted(w.reshape(320, 320), device)
            return d

        def temporal(p):
            return with_projp({"norm": (f32(p + ".norm.weight"), f32(p + ".norm.bias")), "proj_in": lin(p + ".proj_in"),
                               "blk": tblock(p + ".transformer_blocks.0", False), "proj_out": lin(p + ".proj_out")}, p)

        P["time0"], P["time2"] = lin("time_embed.0"), lin("time_embed.2")
        if self.fs_condition:
            P["fps0"], P["fps2"] = lin("fps_embedding.0"), lin("fps_embedding.2")
        if self.addition_attention:
            P["init_attn"] = temporal("init_attn.0")

        def pack_block(prefix, blk):
            out = []
            for j, (kind, a) in enumerate(blk):
                p = f"{prefix}.{j}"
                if kind == "conv_in":
                    d = {"conv": PackedWeight.conv3x3(self._p(p + ".weight"), self._p(p + ".bias"), device)}
                    if a["cin"] <= 8 and os.environ.get("DC_CONVIN_IM2COL", "1") != "0":
                        # 8 input channels: gather the 9 taps into ONE 128-wide K (72 real) and run a plain GEMM - the implicit
                        # conv spends nine K tiles of 64 on them (ops.im2col3x3_c8)
                        d["lin8"] = PackedWeight.conv3x3_c8_as_linear(self._p(p + ".weight"), self._p(p + ".bias"), device)
                    out.append(d)
                elif kind == "res":
                    d = {"gn1": (f32(p + ".in_layers.0.weight"), f32(p + ".in_layers.0.bias")),
                         "conv1": PackedWeight.conv3x3(self._p(p + ".in_layers.2.weight"), self._p(p + ".in_layers.2.bias"), device),
                         "emb": lin(p + ".emb_layers.1"),
                         "gn2": (f32(p + ".out_layers.0.weight"), f32(p + ".out_layers.0.bias")),
                         "conv2": PackedWeight.conv3x3(self._p(p + ".out_layers.3.weight"), self._p(p + ".out_layers.3.bias"), device)}
                    if a["cin"] != a["cout"]:
                        d["skip"] = lin(p + ".skip_connection")
                    if a["tconv"]:
                        d["tc"] = []
                        for i, ci in ((1, 2), (2, 3), (3, 3), (4, 3)):
                            q = f"{p}.temopral_conv.conv{i}"
                            d["tc"].append(((f32(q + ".0.weight"), f32(q + ".0.bias")),
                                            PackedWeight.tconv3(self._p(f"{q}.{ci}.weight"), self._p(f"{q}.{ci}.bias"), device)))
                    out.append(d)
                elif kind == "spatial":
                    out.append(with_projp({"norm": (f32(p + ".norm.weight"), f32(p + ".norm.bias")), "proj_in": lin(p + ".proj_in"),
                                           "blk": tblock(p + ".transformer_blocks.0", True), "proj_out": lin(p + ".proj_out")}, p))
                elif kind == "temporal":
                    out.append(temporal(p))
                elif kind == "down":
                    out.append({"conv": PackedWeight.conv3x3(self._p(p + ".op.weight"), self._p(p + ".op.bias"), device)})
                elif kind == "up":
                    out.append({"conv": PackedWeight.conv3x3(self._p(p + ".conv.weight"), self._p(p + ".conv.bias"), device)})
            return out

        down_path, middle, up_path = self._layout
        P["in"] = [pack_block(f"input_blocks.{i}", b) for i, b in enumerate(down_path)]
        P["mid"] = pack_block("middle_block", middle)
        P["out"] = [pack_block(f"output_blocks.{i}", b) for i, b in enumerate(up_path)]
        P["out_gn"] = (f32("out.0.weight"), f32("out.0.bias"))
        P["out_conv"] = PackedWeight.conv3x3(self._p("out.2.weight"), self._p("out.2.bias"), device, n_align=4)
        P["device"] = device
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1
        P["gen"] = self._pack_gen              # a hoisted context K/V is only valid for the table it was projected with
        return P

    def packed(self, device):
        if self._packed is None or self._packed["device"] != device:
            self._packed = self._pack(device)
        return self._packed

    # ------------------------------------------------------------------ layer recipes on rows
    def _gn(self, x, wb, tag, *, n_inst, rpi, eps, silu):
        y = self._arena.get(tag, x.shape[0], x.shape[1], device=x.device)
        return ops.groupnorm(x, y, wb[0], wb[1], groups=32, n_inst=n_inst, rows_per_inst=rpi, eps=eps, silu=silu)

    def _ln(self, x, wb, tag):
        y = self._arena.get(tag, x.shape[0], x.shape[1], device=x.device)
        return ops.layernorm(x, y, wb[0], wb[1], 1e-5)

    def _res(self, W, x, g, out_tag, out=None):
        A = self._arena
        M, dev = x.shape[0], x.device
        cout = W["conv1"].N
        conv = dict(IH=g["H"], IW=g["W"], OH=g["H"], OW=g["W"], stride=1, pad=1, ups=0)
        emb_out = A.get("emb_out", g["B"], cout, torch.float32, dev)
        ops.gemv_small(g["emb"], W["emb"], emb_out, act_in=1)
        h = self._gn(x, W["gn1"], "gn", n_inst=g["F"], rpi=g["HW"], eps=1e-5, silu=True)
        h1 = ops.gemm(h, W["conv1"], A.get("res_h1", M, cout, device=dev), conv=conv, rowvec=emb_out,
                      rows_per_vec=g["T"] * g["HW"])
        h = self._gn(h1, W["gn2"], "gn", n_inst=g["F"], rpi=g["HW"], eps=1e-5, silu=True)
        skip = x
        if "skip" in W:
            skip = ops.gemm(x, W["skip"], A.get("res_skip", M, cout, device=dev))
        has_tc = "tc" in W
        final = (lambda: out if out is not None else A.get(out_tag, M, cout, device=dev))
        h2 = ops.gemm(h, W["conv2"], A.get("res_h2", M, cout, device=dev) if has_tc else final(), conv=conv,
                      residual=skip)
        if not has_tc:
            return h2
        tc = dict(T=g["T"], HW=g["HW"])
        r = h2
        fused_tc = (_TC_FUSED and cout in _LN_FUSED_K and g["T"] == 16 and g["HW"] % 8 == 0 and M >= _FUSED_MIN_ROWS
                    and (g["T"] * g["HW"]) % 128 == 0)
        for i, (gnw, cw) in enumerate(W["tc"]):
            last = i == 3
            dst = final() if last else A.get("res_ta" if i % 2 == 0 else "res_tb", M, cout, device=dev)
            if fused_tc:
                # levels 0 / 1: GroupNorm statistics pass, then normalise + SiLU + the three taps + bias (+ identity) in one kernel
                st = A.get("gn_stats", g["B"] * 32 * 2, 1, torch.float32, dev)
                ops.groupnorm_stats(r, st, groups=32, n_inst=g["B"], rows_per_inst=g["T"] * g["HW"], eps=1e-5)
                r = ops.gn_silu_tconv3(r, gnw[0], gnw[1], st, cw, dst, B=g["B"], T=16, HW=g["HW"],
                                           residual=h2 if last else None)
                continue
            n = self._gn(r, gnw, "gn", n_inst=g["B"], rpi=g["T"] * g["HW"], eps=1e-5, silu=True)
            r = ops.gemm(n, cw, dst, tconv=tc, residual=h2 if last else None)
        return r

    def _ln_linear(self, h, ln, pw, out):
        """Linear(LayerNorm(h)): at dim 320 / 640 and level-0 / level-1 row counts one kernel with the LayerNorm in registers (the
        normalised copy never reaches HBM), otherwise LayerNorm kernel + GEMM."""
        if _LN_FUSED and pw.K in _LN_FUSED_K and pw.N % 32 == 0 and h.shape[0] >= _FUSED_MIN_ROWS:
            return ops.ln_linear(h, pw, out, ln=ln, ln_eps=1e-5)
        return ops.gemm(self._ln(h, ln, "ln"), pw, out)

    def _lin_res(self, x, pw, res, out):
        """out = res + Linear(x) (attention out-projections, proj_out): at dim 320 / 640 and level-0 / level-1 row counts the
        X-stationary kernel (HBM-bound on x + res + out), otherwise the tile GEMM with the residual through its LDS ring."""
        if _LR_FUSED and pw.K in _LN_FUSED_K and pw.N % 32 == 0 and x.shape[0] >= _FUSED_MIN_ROWS:
            return ops.linear_residual(x, pw, res, out)
        return ops.gemm(x, pw, out, residual=res)

    def _gn_linear(self, x, gnw, pw, out, *, n_inst, rpi):
        """Linear(GroupNorm(x)) (eps 1e-6, no activation: the transformers' norm -> proj_in): at dim 320 / 640 and level-0 / level-1
        row counts the statistics pass plus one kernel that normalises in registers, otherwise GroupNorm + GEMM."""
        if _LN_FUSED and pw.K in _LN_FUSED_K and pw.N % 32 == 0 and x.shape[0] >= _FUSED_MIN_ROWS and rpi % 128 == 0:
            st = self._arena.get("gn_stats", n_inst * 32 * 2, 1, torch.float32, x.device)
            ops.groupnorm_stats(x, st, groups=32, n_inst=n_inst, rows_per_inst=rpi, eps=1e-6)
            return ops.gn_linear(x, gnw[0], gnw[1], st, pw, out, groups=32, rows_per_inst=rpi)
        return ops.gemm(self._gn(x, gnw, "gn", n_inst=n_inst, rpi=rpi, eps=1e-6, silu=False), pw, out)

    def _attn_self_spatial(self, Wa, ln, h, g, heads):
        A = self._arena
        M, dev, Cc = h.shape[0], h.device, heads * 64
        qkv = self._ln_linear(h, ln, Wa["qkv"], A.get("qkv", M, 3 * Cc, device=dev))
        att = A.get("att", M, Cc, device=dev)
        ops.flash_attn(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], att, batch=g["F"], heads=heads, Lq=g["HW"],
                       Lk=g["HW"], scale=0.125)
        return self._lin_res(att, Wa["out"], h, h)

    def _attn_self_temporal(self, Wa, ln, h, g, heads):
        A = self._arena
        M, dev, Cc = h.shape[0], h.device, heads * 64
        att = A.get("att", M, Cc, device=dev)
        if _TA_FUSED and Cc in _TA_FUSED_C and g["T"] == 16 and g["HW"] % 8 == 0 and M >= _FUSED_MIN_ROWS:
            # levels 0 / 1: LayerNorm, q/k/v and the attention over the 16 frames in one kernel - no [M, 3 C] qkv tensor
            ops.ln_qkv_temporal_attn(h, ln, Wa["qkv"], att, B=g["B"], T=16, HW=g["HW"], scale=0.125)
            return self._lin_res(att, Wa["out"], h, h)
        qkv = self._ln_linear(h, ln, Wa["qkv"], A.get("qkv", M, 3 * Cc, device=dev))
        ops.temporal_attn(qkv, att, B=g["B"], T=g["T"], HW=g["HW"], heads=heads, scale=0.125)
        return self._lin_res(att, Wa["out"], h, h)

    def _ff(self, Wb, h, tag="ln"):
        A = self._arena
        if "ff2p" in Wb and _FF_FUSED and h.shape[0] >= _FUSED_MIN_ROWS:
            # dim 320: x = ff(norm3(x)) + x in ONE kernel - LayerNorm in registers, ff1 -> GEGLU -> ff2, residual add; the
            # normalised copy and the [rows, 1280] intermediate never reach HBM
            return ops.ff_geglu_fused320(h, Wb["ff1"], Wb["ff2p"], Wb["ff2"].bias, h, residual=h, ln=Wb["norm3"], ln_eps=1e-5)
        n = self._ln(h, Wb["norm3"], tag)
        mid = ops.gemm(n, Wb["ff1"], A.get("ffmid", h.shape[0], Wb["ff1"].N // 2, device=h.device), geglu=True)
        return ops.gemm(mid, Wb["ff2"], h, residual=h)

    def _ff_proj(self, W, h, x, out):
        """the tail of a transformer: h = h + ff(norm3(h)); return x + proj_out(h) - one kernel at dim 320 / level-0 rows"""
        B_ = W["blk"]
        if "proj_p" in W and _FF_FUSED and _FFP_FUSED and h.shape[0] >= _FUSED_MIN_ROWS:
            return ops.ff_geglu_proj_fused320(h, B_["ff1"], B_["ff2p"], B_["ff2"].bias, W["proj_p"], W["proj_out"].bias, x, out,
                                              ln=B_["norm3"], ln_eps=1e-5)
        h = self._ff(B_, h)
        return self._lin_res(h, W["proj_out"], x, out)

    def _spatial_pre(self, W, x, g, heads):
        """SpatialTransformer up to and including the self-attention residual: everything that does not see the
        context (identical for all guidance branches of one latent)."""
        A = self._arena
        M, dev, Cc = x.shape[0], x.device, x.shape[1]
        h = self._gn_linear(x, W["norm"], W["proj_in"], A.get("tr_h", M, Cc, device=dev), n_inst=g["F"], rpi=g["HW"])
        B_ = W["blk"]
        return self._attn_self_spatial(B_["attn1"], B_["norm1"], h, g, heads)

    def _context_kv(self, B_, ctx, tag):
        """k/v (text) and k_ip/v_ip (image) projections of the per-frame context rows: one GEMM [F*Lc, 4C] (attention.py
        :128-136). The context does not change across the DDIM steps: the sampler path computes these once per run
        (precompute_context_kv), outside the captured step."""
        return ops.gemm(ctx, B_["kv_ctx"], self._arena.get(tag, ctx.shape[0], B_["kv_ctx"].N, device=ctx.device))

    def precompute_context_kv(self, ctx_rows):
        """All cross-attention K/V projections of a run's (step-invariant) context rows -> {block name: kv} + the generation
        of the packed weight table they were computed with (a repack in between - load_state_dict - invalidates them: the
        forward then projects the context itself, as the reference does every step)."""
        Wt = self.packed(ctx_rows.device)
        down_path, middle, up_path = self._layout
        out = {"__gen__": Wt["gen"]}
        for name, blks, Ws in (("in", down_path, Wt["in"]), ("mid", [middle], [Wt["mid"]]), ("out", up_path, Wt["out"])):
            for i, (blk, Wb) in enumerate(zip(blks, Ws)):
                for j, ((kind, a), W) in enumerate(zip(blk, Wb)):
                    if kind == "spatial":
                        out[W["blk"]["key"]] = self._context_kv(W["blk"], ctx_rows, f"kvctx.{name}{i}.{j}")
        return out

    def _spatial_post(self, W, x, h, g, heads, out_tag, out=None):
        """dual cross-attention (shared q; text keys, then image keys accumulated with the image scale), FF, proj_out."""
        A = self._arena
        M, dev, Cc = x.shape[0], x.device, x.shape[1]
        B_ = W["blk"]
        q = self._ln_linear(h, B_["norm2"], B_["q2"], A.get("q2", M, Cc, device=dev))
        kv = None if g.get("ctx_kv") is None else g["ctx_kv"].get(B_["key"])
        if kv is None:
            kv = self._context_kv(B_, g["ctx"], "kvctx")       # rows [F * Lc, 4C], Lc = n_text + L_img
        att = A.get("att", M, Cc, device=dev)
        Lc, nt = g["Lc"], g["n_text"]
        if self.image_cross_attention and Lc > nt:
            # text keys and image keys, two softmaxes, one launch: out = attn_text + ip_scale * attn_image
            ops.cross_attn_dual(q, kv[:, :Cc], kv[:, Cc:2 * Cc], kv[nt:, 2 * Cc:3 * Cc], kv[nt:, 3 * Cc:], att, batch=g["F"],
                                heads=heads, Lq=g["HW"], Lk=nt, Lk2=Lc - nt, scale=0.125, scale2=B_["ip_scale"], kv_bstride=Lc)
        else:
            ops.flash_attn(q, kv[:, :Cc], kv[:, Cc:2 * Cc], att, batch=g["F"], heads=heads, Lq=g["HW"], Lk=nt, scale=0.125,
                           kv_bstride=Lc)
        h = self._lin_res(att, B_["out2"], h, h)
        return self._ff_proj(W, h, x, out if out is not None else A.get(out_tag, M, Cc, device=dev))

    def _spatial(self, W, x, g, heads, out_tag, out=None):
        return self._spatial_post(W, x, self._spatial_pre(W, x, g, heads), g, heads, out_tag, out=out)

    def _temporal(self, W, x, g, heads, out_tag, out=None):
        A = self._arena
        M, dev, Cc = x.shape[0], x.device, x.shape[1]
        inner = W["proj_in"].N
        h = self._gn_linear(x, W["norm"], W["proj_in"], A.get("tr_h", M, inner, device=dev), n_inst=g["B"],
                            rpi=g["T"] * g["HW"])
        B_ = W["blk"]
        h = self._attn_self_temporal(B_["attn1"], B_["norm1"], h, g, heads)
        h = self._attn_self_temporal(B_["attn2"], B_["norm2"], h, g, heads)
        return self._ff_proj(W, h, x, out if out is not None else A.get(out_tag, M, Cc, device=dev))

    def _run_block(self, blk, Wb, h, g, tag, final=None):
        """`final(rows, cols)` -> the tensor the LAST layer of the block must write (a column range of a skip-concat
        buffer); None = a scratch buffer of its own."""
        A = self._arena
        for j, ((kind, a), W) in enumerate(zip(blk, Wb)):
            out_tag = f"{tag}.{j}"
            dev = h.device
            last = final is not None and j == len(blk) - 1
            if kind == "conv_in":
                conv = dict(IH=g["H"], IW=g["W"], OH=g["H"], OW=g["W"], stride=1, pad=1, ups=0)
                dst = final(h.shape[0], a["cout"]) if last else A.get(out_tag, h.shape[0], a["cout"], device=dev)
                if "lin8" in W:
                    cols = ops.im2col3x3_c8(h, A.get(out_tag + ".im2col", h.shape[0], 128, device=dev), n_img=h.shape[0] // (g["H"] * g["W"]),
                                            H=g["H"], W=g["W"])
                    h = ops.gemm(cols, W["lin8"], dst)
                else:
                    h = ops.gemm(h, W["conv"], dst, conv=conv)
            elif kind == "res":
                h = self._res(W, h, g, out_tag, out=final(h.shape[0], a["cout"]) if last else None)
            elif kind == "spatial":
                h = self._spatial(W, h, g, a["heads"], out_tag, out=final(h.shape[0], a["ch"]) if last else None)
            elif kind == "temporal":
                h = self._temporal(W, h, g, a["heads"], out_tag, out=final(h.shape[0], a["ch"]) if last else None)
            elif kind == "down":
                OH, OW = (g["H"] + 1) // 2, (g["W"] + 1) // 2       # 3x3 s2 p1: floor((H-1)/2)+1
                conv = dict(IH=g["H"], IW=g["W"], OH=OH, OW=OW, stride=2, pad=1, ups=0)
                rows = g["F"] * OH * OW
                h = ops.gemm(h, W["conv"], final(rows, a["ch"]) if last else A.get(out_tag, rows, a["ch"], device=dev), conv=conv)
                g["H"], g["W"], g["HW"] = OH, OW, OH * OW
            elif kind == "up":
                OH, OW = g["H"] * 2, g["W"] * 2
                conv = dict(IH=g["H"], IW=g["W"], OH=OH, OW=OW, stride=1, pad=1, ups=1)
                rows = g["F"] * OH * OW
                h = ops.gemm(h, W["conv"], final(rows, a["ch"]) if last else A.get(out_tag, rows, a["ch"], device=dev), conv=conv)
                g["H"], g["W"], g["HW"] = OH, OW, OH * OW
        return h

    # ------------------------------------------------------------------ forward on rows
    def _replicate(self, src, tag, nrep, dst=None):
        """rows [M, C] -> [nrep*M, C] (the guidance branches start from identical activations)"""
        if dst is None:
            dst = self._arena.get(tag, nrep * src.shape[0], src.shape[1], device=src.device)
        for k in range(nrep):
            ops.copy2d(src, dst[k * src.shape[0]:(k + 1) * src.shape[0]])
        return dst

    def forward_rows(self, xrows, t_table, ctx_rows, *, B, T, H, W, Lc, n_text=77, fs_table=None, t_index=None,
                     shared_prefix=1, ctx_kv=None):
        """xrows: bf16 [B*T*H*W, 64] (latent+concat channels, zero padded); t_table int64 [*, B] (row selected by
        the device counter t_index, or row 0); ctx_rows bf16 [B*T*Lc, context_dim]; fs_table int64 [B];
        ctx_kv: precompute_context_kv(ctx_rows) or None (projected here).
        Returns fp32 rows [B*T*H*W, 4] (channels-last model output)."""
        dev = xrows.device
        Wt = self.packed(dev)
        A = self._arena
        mc = self.model_channels
        # timestep / fps embedding MLPs (openaimodel3d.py:550-577); emb is per clip b (repeat over t is implicit)
        temb = A.get("t_sin", B, mc, torch.float32, dev)
        ops.timestep_embedding(t_table, temb, mc, t_index=t_index, t_stride=B)
        hid = A.get("emb_hid", B, 4 * mc, torch.float32, dev)
        emb = A.get("emb", B, 4 * mc, torch.float32, dev)
        ops.gemv_small(temb, Wt["time0"], hid, act_out=1)
        ops.gemv_small(hid, Wt["time2"], emb)
        if self.fs_condition:
            femb = A.get("fs_sin", B, mc, torch.float32, dev)
            ops.timestep_embedding(fs_table, femb, mc)
            ops.gemv_small(femb, Wt["fps0"], hid, act_out=1)
            ops.gemv_small(hid, Wt["fps2"], emb, accumulate=True)
        if ctx_kv is not None and ctx_kv.get("__gen__") != Wt["gen"]:
            ctx_kv = None                      # projected with weights that have been repacked since: project here
        g = dict(B=B, T=T, F=B * T, H=H, W=W, HW=H * W, emb=emb, ctx=ctx_rows, Lc=Lc, n_text=n_text, ctx_kv=ctx_kv)
        down_path, middle, up_path = self._layout
        n_up = len(up_path)

        # Skip connections without a concat copy (reference: torch.cat([h, hs.pop()], dim=1), openaimodel3d.py:596):
        # up block u reads ONE buffer cat{u} = [h | skip]; the down block that produces the skip and the block that
        # produces h each write their column range of it directly (every kernel takes a row stride).
        cat_rows = {}

        def cat_view(u, rows, part):
            ch, cs = self._cat_plan[u]
            if cat_rows.setdefault(u, rows) != rows:
                raise ValueError("skip / decoder resolution mismatch (the latent height and width must divide by 8)")
            buf = A.get(f"cat{u}", rows, ch + cs, device=dev)
            return buf[:, :ch] if part == "h" else buf[:, ch:]

        def skip_dst(d):                       # output of down block d = the skip half of up block n-1-d's input
            return lambda rows, cols: cat_view(n_up - 1 - d, rows, "skip")

        h = xrows
        first = 0
        nrep = shared_prefix
        if nrep > 1 and len(down_path) > 1 and [k for k, _ in down_path[1]][:2] == ["res", "spatial"]:
            # The `nrep` guidance branches share latent, c_concat, timestep and fs: everything before the first
            # cross-attention (conv_in, init_attn, the first ResBlock, the first SpatialTransformer's self-attention)
            # is computed ONCE on B/nrep clips and replicated; results are bit-identical to the full batch.
            B1 = B // nrep
            g1 = dict(g, B=B1, F=B1 * T, emb=emb[:B1])      # the branches share timestep and fs: rows 0..B1 of emb
            h = self._run_block(down_path[0], Wt["in"][0], h[:B1 * T * H * W], g1, "in0")
            if self.addition_attention:
                h = self._temporal(Wt["init_attn"], h, g1, 8, "init_attn")
            self._replicate(h, None, nrep, dst=skip_dst(0)(nrep * h.shape[0], h.shape[1]))
            blk, Wb = down_path[1], Wt["in"][1]
            r = self._res(Wb[0], h, g1, "in1.0")
            hh = self._spatial_pre(Wb[1], r, g1, blk[1][1]["heads"])
            r_full = self._replicate(r, "in1.0.rep", nrep)
            h_full = self._replicate(hh, "tr_h.rep", nrep)
            last = len(blk) - 1
            fin = skip_dst(1)
            h = self._spatial_post(Wb[1], r_full, h_full, g, blk[1][1]["heads"], "in1.1",
                                   out=fin(r_full.shape[0], r_full.shape[1]) if last == 1 else None)
            for j in range(2, len(blk)):
                kind, a = blk[j]
                assert kind == "temporal"
                h = self._temporal(Wb[j], h, g, a["heads"], f"in1.{j}", out=fin(h.shape[0], a["ch"]) if j == last else None)
            first = 2
        for i, blk in enumerate(down_path):
            if i < first:
                continue
            if i == 0 and self.addition_attention:
                h = self._run_block(blk, Wt["in"][i], h, g, f"in{i}")
                h = self._temporal(Wt["init_attn"], h, g, 8, "init_attn", out=skip_dst(0)(h.shape[0], h.shape[1]))
            else:
                h = self._run_block(blk, Wt["in"][i], h, g, f"in{i}", final=skip_dst(i))
        h = self._run_block(middle, Wt["mid"], h, g, "mid", final=lambda rows, cols: cat_view(0, rows, "h"))
        for i, blk in enumerate(up_path):
            ch, cs = self._cat_plan[i]
            if cat_rows.get(i) != h.shape[0]:
                raise ValueError("skip / decoder resolution mismatch (the latent height and width must divide by 8)")
            cat = A.get(f"cat{i}", h.shape[0], ch + cs, device=dev)
            assert h.data_ptr() == cat.data_ptr() and h.shape[1] == ch
            nxt = (lambda rows, cols, u=i + 1: cat_view(u, rows, "h")) if i + 1 < n_up else None
            h = self._run_block(blk, Wt["out"][i], cat, g, f"out{i}", final=nxt)
        n = self._gn(h, Wt["out_gn"], "gn", n_inst=g["F"], rpi=g["HW"], eps=1e-5, silu=True)
        y = A.get("unet_out", h.shape[0], Wt["out_conv"].N, torch.float32, dev)
        conv = dict(IH=g["H"], IW=g["W"], OH=g["H"], OW=g["W"], stride=1, pad=1, ups=0)
        return ops.gemm(n, Wt["out_conv"], y, conv=conv)

    def build_context_rows(self, context, B, T, tag="ctx"):
        """context fp32 [B, L, D] -> per-frame bf16 rows [B*T*Lc, D] (openaimodel3d.py:555-562)."""
        dev = context.device
        L, D = context.shape[1], context.shape[2]
        context = context.to(torch.float32).contiguous()
        if L == 77 + T * 16:
            Lc = 77 + 16
            out = self._arena.get(tag, B * T * Lc, D, device=dev)
            ops.build_context(context, out, B=B, T=T, n_text=77, L=16, D=D)
        else:
            # every frame sees the same tokens (context.repeat_interleave(t)); image tokens = those past 77
            Lc = L
            n_text = min(77, L)
            rep = context.repeat_interleave(T, dim=0).contiguous()
            out = self._arena.get(tag, B * T * Lc, D, device=dev)
            ops.build_context(rep, out, B=B * T, T=1, n_text=n_text, L=L - n_text, D=D)
        return out, Lc

    def forward_pair(self, x, cc, timesteps, context, fs=None, **kwargs):
        """forward() with the channel concat of DiffusionWrapper ('hybrid', ddpm3d.py:1254-1258) fused into the
        input packing kernel: x [B,Cx,T,h,w] and cc [B,Cc,T,h,w] (or None), Cx + Cc == in_channels."""
        if not x.is_cuda:
            raise RuntimeError("UNetModel runs on the HIP path only: move the model and inputs to the GPU "
                               "(there is no CPU fallback)")
        B, Cx, T, H, W = x.shape
        Cc = 0 if cc is None else cc.shape[1]
        assert Cx + Cc == self.in_channels
        dev = x.device
        xr = self._arena.get("x_rows", B * T * H * W, C_IN_PAD, device=dev)
        ops.pack_latent(x.to(torch.float32).contiguous(), None if cc is None else cc.to(torch.float32).contiguous(),
                        xr, B=B, Cx=Cx, Cc=Cc, T=T, HW=H * W)
        ctx_rows, Lc = self.build_context_rows(context, B, T)
        t_table = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        if fs is None:
            fs = torch.full((B,), self.default_fs, dtype=torch.int64, device=dev)
        fs_table = fs.to(device=dev, dtype=torch.int64).contiguous()
        y = self.forward_rows(xr, t_table, ctx_rows, B=B, T=T, H=H, W=W, Lc=Lc, n_text=min(77, Lc), fs_table=fs_table)
        out = torch.empty((B, self.out_channels, T, H, W), dtype=torch.float32, device=dev)
        ops.rows_to_nchw(y, out, N=B, Cc=self.out_channels, HW=T * H * W)
        return out

    def forward(self, x, timesteps, context=None, features_adapter=None, fs=None, **kwargs):
        """x [B, in_channels, T, h, w] -> [B, out_channels, T, h, w] (fp32), reference signature :548."""
        if features_adapter is not None:
            raise NotImplementedError("features_adapter is not part of the DynamiCrafter inference path")
        return self.forward_pair(x, None, timesteps, context, fs=fs)
