"""AutoencoderKL (SD-VAE f8) encode/decode as HIP kernel sequences on channels-last rows.

Drop-in for lvdm/models/autoencoder.py:13-107 (+ Encoder/Decoder/ResnetBlock/AttnBlock/Down/Upsample of
lvdm/modules/networks/ae_modules.py): same constructor keywords (`first_stage_config.params`), same
`encode(x) -> posterior`, `decode(z)`, same state_dict keys (`encoder.*`, `decoder.*`, `quant_conv.*`,
`post_quant_conv.*`). All convolutions are dc_gemm_conv implicit GEMMs, GroupNorm(eps 1e-6)+swish is
dc_groupnorm, and the single-head d=512 mid attention is two MFMA GEMMs around a row softmax per frame.
"""
import torch
import torch.nn as nn

from ... import ops
from ...ops import PackedWeight
from ...param_tree import attach_params
from ..distributions import DiagonalGaussianDistribution

_BF16 = torch.bfloat16
PADC = 64


def _ae_shapes(dd, embed_dim):
    t = {}
    ch, mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    zc = dd["z_channels"]

    def affine(p, c):
        t[p + ".weight"] = (c,); t[p + ".bias"] = (c,)

    def conv(p, i, o, k):
        t[p + ".weight"] = (o, i, k, k); t[p + ".bias"] = (o,)

    def res(p, i, o):
        affine(p + ".norm1", i); conv(p + ".conv1", i, o, 3)
        affine(p + ".norm2", o); conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".nin_shortcut", i, o, 1)

    def attn(p, c):
        affine(p + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{p}.{n}", c, c, 1)

    n = len(mult)
    conv("encoder.conv_in", dd["in_channels"], ch, 3)
    cur = ch
    for lvl in range(n):
        for i in range(nrb):
            res(f"encoder.down.{lvl}.block.{i}", cur, ch * mult[lvl]); cur = ch * mult[lvl]
        if lvl != n - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", cur, cur, 3)
    res("encoder.mid.block_1", cur, cur); attn("encoder.mid.attn_1", cur); res("encoder.mid.block_2", cur, cur)
    affine("encoder.norm_out", cur)
    conv("encoder.conv_out", cur, 2 * zc if dd.get("double_z", True) else zc, 3)
    cur = ch * mult[-1]
    conv("decoder.conv_in", zc, cur, 3)
    res("decoder.mid.block_1", cur, cur); attn("decoder.mid.attn_1", cur); res("decoder.mid.block_2", cur, cur)
    for lvl in reversed(range(n)):
        for i in range(nrb + 1):
            res(f"decoder.up.{lvl}.block.{i}", cur, ch * mult[lvl]); cur = ch * mult[lvl]
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", cur, cur, 3)
    affine("decoder.norm_out", cur)
    conv("decoder.conv_out", cur, dd["out_ch"], 3)
    conv("quant_conv", 2 * zc, 2 * embed_dim, 1)
    conv("post_quant_conv", embed_dim, zc, 1)
    return t


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None):
        super().__init__()
        dd = dict(ddconfig)
        assert dd["double_z"]
        if dd.get("attn_resolutions"):
            raise NotImplementedError("attention inside AE levels is not used by the released configs")
        if dd["ch"] % 64 != 0:
            raise NotImplementedError("AE base width must be a multiple of 64 (one MFMA K slice)")
        self.ddconfig = dd
        self.embed_dim = embed_dim
        self.image_key = image_key
        if monitor is not None:
            self.monitor = monitor
        attach_params(self, _ae_shapes(dd, embed_dim))
        self._packed = None
        self._arena = ops.Arena()
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_packed", None))
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu")
            sd = sd.get("state_dict", sd)
            sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
            self.load_state_dict(sd, strict=False)

    # ---- plumbing -------------------------------------------------------------------------------------
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

    def _buf(self, tag, rows, cols, dtype=_BF16, device=None, zero=False):
        return self._arena.get(tag, rows, cols, dtype, device, zero=zero)

    def _pack(self, device):
        f32 = lambda n: self._p(n).detach().to(device=device, dtype=torch.float32).contiguous()
        c3 = lambda n, **kw: PackedWeight.conv3x3(self._p(n + ".weight"), self._p(n + ".bias"), device, **kw)
        c1 = lambda n, **kw: PackedWeight.linear(self._p(n + ".weight"), self._p(n + ".bias"), device, **kw)
        gn = lambda n: (f32(n + ".weight"), f32(n + ".bias"))

        def res(p):
            d = {"n1": gn(p + ".norm1"), "c1": c3(p + ".conv1"), "n2": gn(p + ".norm2"), "c2": c3(p + ".conv2")}
            if self._has(p + ".nin_shortcut.weight"):
                d["nin"] = c1(p + ".nin_shortcut")
            return d

        def attn(p):
            return {"n": gn(p + ".norm"), "q": c1(p + ".q"), "k": c1(p + ".k"), "v": c1(p + ".v"), "o": c1(p + ".proj_out")}

        dd = self.ddconfig
        n, nrb = len(dd["ch_mult"]), dd["num_res_blocks"]
        P = {"device": device}
        P["enc_in"] = c3("encoder.conv_in")
        P["enc_down"] = []
        for lvl in range(n):
            P["enc_down"].append({"blocks": [res(f"encoder.down.{lvl}.block.{i}") for i in range(nrb)],
                                  "down": c3(f"encoder.down.{lvl}.downsample.conv") if lvl != n - 1 else None})
        P["enc_mid"] = (res("encoder.mid.block_1"), attn("encoder.mid.attn_1"), res("encoder.mid.block_2"))
        P["enc_norm"] = gn("encoder.norm_out")
        P["enc_out"] = c3("encoder.conv_out", n_align=PADC)      # 8 -> 64 zero-padded so quant_conv sees clean K
        P["quant"] = c1("quant_conv", n_align=PADC)
        P["post_quant"] = c1("post_quant_conv", n_align=PADC)
        P["dec_in"] = c3("decoder.conv_in")
        P["dec_mid"] = (res("decoder.mid.block_1"), attn("decoder.mid.attn_1"), res("decoder.mid.block_2"))
        P["dec_up"] = {}
        for lvl in range(n):
            P["dec_up"][lvl] = {"blocks": [res(f"decoder.up.{lvl}.block.{i}") for i in range(nrb + 1)],
                                "up": c3(f"decoder.up.{lvl}.upsample.conv") if lvl != 0 else None}
        P["dec_norm"] = gn("decoder.norm_out")
        P["dec_out"] = c3("decoder.conv_out", n_align=4)
        return P

    def packed(self, device):
        if self._packed is None or self._packed["device"] != device:
            self._packed = self._pack(device)
        return self._packed

    # ---- layer recipes on rows [N*H*W, C] ---------------------------------------------------------------
    def _gn_act(self, x, wb, N, HW, act=True):
        y = self._buf("gn", x.shape[0], x.shape[1], device=x.device)
        return ops.groupnorm(x, y, wb[0], wb[1], groups=32, n_inst=N, rows_per_inst=HW, eps=1e-6, silu=act)

    def _res(self, W, x, N, H, Wd, tag):
        """ResnetBlock.forward ae_modules.py:190-210"""
        dev, M = x.device, x.shape[0]
        conv = dict(IH=H, IW=Wd, OH=H, OW=Wd, stride=1, pad=1, ups=0)
        co = W["c1"].N
        h = ops.gemm(self._gn_act(x, W["n1"], N, H * Wd), W["c1"], self._buf("res_h", M, co, device=dev), conv=conv)
        skip = x if "nin" not in W else ops.gemm(x, W["nin"], self._buf("res_s", M, co, device=dev))
        return ops.gemm(self._gn_act(h, W["n2"], N, H * Wd), W["c2"], self._buf(tag, M, co, device=dev), conv=conv,
                        residual=skip)

    def _attn(self, W, x, N, HW, tag):
        """AttnBlock.forward ae_modules.py:53-78: per frame softmax(q k^T c^-1/2) v, one head of width c."""
        dev, M, Cc = x.device, x.shape[0], x.shape[1]
        hn = self._gn_act(x, W["n"], N, HW, act=False)
        q = ops.gemm(hn, W["q"], self._buf("at_q", M, Cc, device=dev))
        v = ops.gemm(hn, W["v"], self._buf("at_v", M, Cc, device=dev))
        kpad = (HW + 127) // 128 * 128
        kbuf = self._buf("at_k", (N - 1) * HW + kpad, Cc, device=dev, zero=True)   # tail rows stay zero
        ops.gemm(hn, W["k"], kbuf[:M])
        kp = (HW + 63) // 64 * 64
        S = self._buf("at_s", HW, HW, torch.float32, dev)
        Pm = self._buf("at_p", HW, kp, device=dev, zero=True)                          # pad columns stay zero
        vt = self._buf("at_vt", (Cc + 127) // 128 * 128, kp, device=dev, zero=True)
        o = self._buf("at_o", M, Cc, device=dev)
        for f in range(N):
            rows = slice(f * HW, (f + 1) * HW)
            kw = PackedWeight(kbuf[f * HW:f * HW + kpad], None, HW, Cc, Cc, 1)
            ops.gemm(q[rows], kw, S, alpha=float(Cc) ** -0.5)
            ops.softmax_rows(S, Pm[:, :HW])
            ops.transpose(v[rows], vt, rows=HW, cols=Cc)
            vw = PackedWeight(vt, None, Cc, kp, kp, 1)
            ops.gemm(Pm, vw, o[rows])
        return ops.gemm(o, W["o"], self._buf(tag, M, Cc, device=dev), residual=x)

    def _encode_rows(self, x):
        """Encoder.forward ae_modules.py:430-463 + quant_conv autoencoder.py:99. x fp32 [N,3,H,W] -> moments rows."""
        P = self.packed(x.device)
        dev = x.device
        N, Cin, H, Wd = x.shape
        dd = self.ddconfig
        n = len(dd["ch_mult"])
        h = self._buf("enc_x", N * H * Wd, PADC, device=dev)
        ops.nchw_to_rows(x.to(torch.float32).contiguous(), h, N=N, Cc=Cin, HW=H * Wd)
        h = ops.gemm(h, P["enc_in"], self._buf("enc_in", N * H * Wd, P["enc_in"].N, device=dev),
                     conv=dict(IH=H, IW=Wd, OH=H, OW=Wd, stride=1, pad=1, ups=0))
        for lvl in range(n):
            L = P["enc_down"][lvl]
            for i, Wb in enumerate(L["blocks"]):
                h = self._res(Wb, h, N, H, Wd, f"enc{lvl}.{i}")
            if L["down"] is not None:
                # Downsample: F.pad (0,1,0,1) then 3x3 stride 2 pad 0 (ae_modules.py:102-106)
                OH, OW = (H + 1 - 3) // 2 + 1, (Wd + 1 - 3) // 2 + 1
                h = ops.gemm(h, L["down"], self._buf(f"encdown{lvl}", N * OH * OW, L["down"].N, device=dev),
                             conv=dict(IH=H, IW=Wd, OH=OH, OW=OW, stride=2, pad=0, ups=0))
                H, Wd = OH, OW
        r1, at, r2 = P["enc_mid"]
        h = self._res(r1, h, N, H, Wd, "encmid1")
        h = self._attn(at, h, N, H * Wd, "encmida")
        h = self._res(r2, h, N, H, Wd, "encmid2")
        h = self._gn_act(h, P["enc_norm"], N, H * Wd)
        h = ops.gemm(h, P["enc_out"], self._buf("enc_out", N * H * Wd, PADC, device=dev),
                     conv=dict(IH=H, IW=Wd, OH=H, OW=Wd, stride=1, pad=1, ups=0))
        mom = ops.gemm(h, P["quant"], self._buf("enc_mom", N * H * Wd, PADC, device=dev))
        return mom, (N, H, Wd)

    def latent_hw(self, H, Wd):
        """spatial size of the latent of an [.., H, W] image: one pad(0,1,0,1) + 3x3 stride-2 conv per level but the last"""
        for _ in range(len(self.ddconfig["ch_mult"]) - 1):
            H, Wd = (H + 1 - 3) // 2 + 1, (Wd + 1 - 3) // 2 + 1
        return H, Wd

    def encode(self, x, **kwargs):
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKL runs on the HIP path only (no CPU fallback)")
        mom, (N, H, Wd) = self._encode_rows(x)
        return DiagonalGaussianDistribution(mom.clone(), zc=self.embed_dim, N=N, H=H, W=Wd)

    def decode(self, z, **kwargs):
        """AutoencoderKL.decode autoencoder.py:104-107; Decoder.forward ae_modules.py:539-578. z fp32 [N,4,h,w]."""
        if not z.is_cuda:
            raise RuntimeError("AutoencoderKL runs on the HIP path only (no CPU fallback)")
        P = self.packed(z.device)
        dev = z.device
        N, zc, H, Wd = z.shape
        n = len(self.ddconfig["ch_mult"])
        h = self._buf("dec_z", N * H * Wd, PADC, device=dev)
        ops.nchw_to_rows(z.to(torch.float32).contiguous(), h, N=N, Cc=zc, HW=H * Wd)
        h = ops.gemm(h, P["post_quant"], self._buf("dec_pq", N * H * Wd, PADC, device=dev))
        h = ops.gemm(h, P["dec_in"], self._buf("dec_in", N * H * Wd, P["dec_in"].N, device=dev),
                     conv=dict(IH=H, IW=Wd, OH=H, OW=Wd, stride=1, pad=1, ups=0))
        r1, at, r2 = P["dec_mid"]
        h = self._res(r1, h, N, H, Wd, "decmid1")
        h = self._attn(at, h, N, H * Wd, "decmida")
        h = self._res(r2, h, N, H, Wd, "decmid2")
        for lvl in reversed(range(n)):
            L = P["dec_up"][lvl]
            for i, Wb in enumerate(L["blocks"]):
                h = self._res(Wb, h, N, H, Wd, f"dec{lvl}.{i}")
            if L["up"] is not None:
                OH, OW = 2 * H, 2 * Wd
                h = ops.gemm(h, L["up"], self._buf(f"decup{lvl}", N * OH * OW, L["up"].N, device=dev),
                             conv=dict(IH=H, IW=Wd, OH=OH, OW=OW, stride=1, pad=1, ups=1))
                H, Wd = OH, OW
        h = self._gn_act(h, P["dec_norm"], N, H * Wd)
        y = ops.gemm(h, P["dec_out"], self._buf("dec_out", N * H * Wd, P["dec_out"].N, device=dev),
                     conv=dict(IH=H, IW=Wd, OH=H, OW=Wd, stride=1, pad=1, ups=0))
        out = torch.empty((N, self.ddconfig["out_ch"], H, Wd), dtype=torch.float32, device=dev)
        ops.rows_to_nchw(y, out, N=N, Cc=self.ddconfig["out_ch"], HW=H * Wd)
        return out

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
