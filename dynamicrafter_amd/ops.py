"""Tensor-level wrappers over the C ABI (include/dcrafter_hip.h).

Every function takes torch CUDA tensors purely as (pointer, stride) carriers and enqueues HIP kernels on the
current torch stream. Activations are channels-last bf16 rows: 2-D tensors [rows, C] with stride (ld, 1).
Nothing here computes with torch.
"""
import ctypes as C
import os
import math

import torch

from . import _hip
from ._hip import DcDdimParams, DcGemmParams, check

_BF16 = torch.bfloat16


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Tracer:
    """Per-launch HIP-event timing + algorithmic work accounting (bench.py's roofline leg). While a Tracer is
    installed (`with Tracer() as tr:`) every heavy op brackets its launch with two hipEvents on the launch stream;
    `summary()` reduces them per kernel family: launches, total ms, FLOPs, algorithmic HBM bytes."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _TRACE
        _TRACE = self
        return self

    def __exit__(self, *a):
        global _TRACE
        _TRACE = None

    def _event(self):
        e = C.c_void_p()
        check(_hip.lib().dc_event_create(C.byref(e)), "dc_event_create")
        return e

    def summary(self):
        l = _hip.lib()
        out = {}
        self.detail = {}
        for name, flops, nbytes, e0, e1, tag in self.records:
            ms = C.c_float()
            check(l.dc_event_elapsed_ms(e0, e1, C.byref(ms)), "dc_event_elapsed_ms")
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1; d["ms"] += ms.value; d["flops"] += flops; d["bytes"] += nbytes
            if tag is not None:
                dd = self.detail.setdefault((name, tag), dict(launches=0, ms=0.0, flops=0.0))
                dd["launches"] += 1; dd["ms"] += ms.value; dd["flops"] += flops
            l.dc_event_destroy(e0); l.dc_event_destroy(e1)
        self.records = []
        return out


_TRACE = None


def _launch(name, flops, nbytes, fn, *args, tag=None):
    tr = _TRACE
    if tr is None:
        check(fn(*args), name)
        return
    l = _hip.lib()
    e0, e1 = tr._event(), tr._event()
    sp = stream_ptr()
    l.dc_event_record(e0, sp)
    check(fn(*args), name)
    l.dc_event_record(e1, sp)
    if name == "dc_gemm_conv":                      # label by the kernel family the dispatcher actually launched
        name = l.dc_gemm_last_variant().decode()
    tr.records.append((name, flops, nbytes, e0, e1, tag))


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _rows(t, name="tensor", dtype=_BF16):
    if t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise ValueError(f"{name}: expected CUDA {dtype} rows [R, C] with unit column stride, got "
                         f"{t.dtype} {tuple(t.shape)} strides {t.stride()} on {t.device}")
    return t


class PackedWeight:
    """Device copy of a Linear / conv weight in the layout dc_gemm_conv consumes: bf16 [n_pad][K]; conv K is
    ordered (64-channel slice, tap, channel); rows zero-padded to a multiple of 128. Derived from the
    nn.Parameter, never serialised."""

    __slots__ = ("w", "bias", "N", "K", "n_pad", "Cin", "taps", "k_real")

    def __init__(self, w, bias, N, K, Cin, taps, k_real=None):
        self.w, self.bias, self.N, self.K, self.Cin, self.taps = w, bias, N, K, Cin, taps
        self.n_pad = w.shape[0]
        self.k_real = k_real        # un-padded reduction length (algorithmic FLOP accounting)

    @staticmethod
    def _finish(w2d, bias, device, Cin, taps, pad_n_to=None):
        N, K = w2d.shape
        n_pad = (N + 127) // 128 * 128
        if pad_n_to is not None:
            n_pad = max(n_pad, pad_n_to)
        wp = torch.zeros((n_pad, K), dtype=_BF16, device=device)
        wp[:N] = w2d.to(device=device, dtype=_BF16)
        b = None if bias is None else bias.detach().to(device=device, dtype=torch.float32).contiguous()
        return PackedWeight(wp, b, N, K, Cin, taps)

    @staticmethod
    def linear(weight, bias, device, n_align=1):
        """nn.Linear / Conv1d(k=1) / Conv2d(k=1) weight [N, K, ...]. n_align pads N (zero rows + zero bias)."""
        w = weight.detach().reshape(weight.shape[0], -1)
        N, K = w.shape
        if K % 64 != 0:
            kp = (K + 63) // 64 * 64
            w = torch.nn.functional.pad(w, (0, kp - K))
        if N % n_align != 0:
            npad = (N + n_align - 1) // n_align * n_align
            w = torch.nn.functional.pad(w, (0, 0, 0, npad - N))
            if bias is not None:
                bias = torch.nn.functional.pad(bias.detach(), (0, npad - N))
        return PackedWeight._finish(w, bias, device, w.shape[1], 1)

    @staticmethod
    def conv3x3(weight, bias, device, n_align=1):
        """nn.Conv2d 3x3 weight [Cout, Cin, 3, 3] -> [Cout][kh][kw][Cin_pad64]."""
        co, ci = weight.shape[0], weight.shape[1]
        w = weight.detach().permute(0, 2, 3, 1)
        cip = (ci + 63) // 64 * 64
        if cip != ci:
            w = torch.nn.functional.pad(w, (0, cip - ci))
        # K order = (64-channel slice, tap, channel-in-slice): the kernel walks all 9 taps of a slice back to back
        w = w.reshape(co, 9, cip // 64, 64).permute(0, 2, 1, 3).reshape(co, 9 * cip)
        if co % n_align != 0:
            npad = (co + n_align - 1) // n_align * n_align
            w = torch.nn.functional.pad(w, (0, 0, 0, npad - co))
            if bias is not None:
                bias = torch.nn.functional.pad(bias.detach(), (0, npad - co))
        pw = PackedWeight._finish(w, bias, device, cip, 9)
        pw.k_real = 9 * ci
        return pw

    @staticmethod
    def conv3x3_c8_as_linear(weight, bias, device):
        """nn.Conv2d 3x3 weight [Cout, Cin <= 8, 3, 3] -> the Linear weight [Cout][128] that goes with ops.im2col3x3_c8:
        k = 8 (kh*3 + kw) + c, zeros elsewhere."""
        co, ci = weight.shape[0], weight.shape[1]
        assert ci <= 8
        w = torch.zeros(co, 16, 8, dtype=weight.dtype)
        w[:, :9, :ci] = weight.detach().permute(0, 2, 3, 1).reshape(co, 9, ci)
        pw = PackedWeight._finish(w.reshape(co, 128), bias, device, 128, 1)
        pw.k_real = 9 * ci
        return pw

    @staticmethod
    def tconv3(weight, bias, device):
        """nn.Conv3d (3,1,1) weight [Cout, Cin, 3, 1, 1] -> [Cout][kt][Cin]."""
        co, ci = weight.shape[0], weight.shape[1]
        assert ci % 64 == 0, "temporal conv width must be a multiple of 64"
        w = weight.detach().reshape(co, ci // 64, 64, 3).permute(0, 1, 3, 2).reshape(co, 3 * ci)   # (slice, tap, channel)
        return PackedWeight._finish(w, bias, device, ci, 3)


def gemm(a, pw, out, *, M=None, residual=None, rowvec=None, rows_per_vec=1, geglu=False, gelu=False, alpha=1.0,
         conv=None, tconv=None):
    """out[M, N] = epilogue(gather(a) @ pw.w[:N].T). `out` dtype bf16 or float32 selects the output type.

    conv  = dict(IH, IW, OH, OW, stride, pad, ups)   -> 3x3 implicit GEMM over frames of a
    tconv = dict(T, HW)                               -> 3-tap temporal conv
    """
    _rows(a, "a")
    out_f32 = out.dtype == torch.float32
    if not out_f32:
        _rows(out, "out")
    p = DcGemmParams()
    p.A, p.W, p.C = a.data_ptr(), pw.w.data_ptr(), out.data_ptr()
    p.bias = 0 if pw.bias is None else pw.bias.data_ptr()
    p.rowvec = 0 if rowvec is None else rowvec.data_ptr()
    p.rowvec_ld = 0 if rowvec is None else rowvec.stride(0)
    p.rows_per_vec = rows_per_vec
    p.residual = 0 if residual is None else _rows(residual, "residual").data_ptr()
    p.ldr = 0 if residual is None else residual.stride(0)
    p.lda, p.ldc = a.stride(0), out.stride(0)
    p.M = out.shape[0] if M is None else M
    p.N, p.K, p.n_pad = pw.N, pw.K, pw.n_pad
    p.Cin = pw.Cin
    p.flags = (_hip.DC_GEMM_OUT_F32 if out_f32 else 0) | (_hip.DC_GEMM_GEGLU if geglu else 0) | \
        (_hip.DC_GEMM_GELU if gelu else 0)
    p.alpha = alpha
    ws = _gemm_workspace(a.device, torch.cuda.current_stream().cuda_stream)
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws.numel()
    if conv is not None:
        p.mode = 1
        p.IH, p.IW, p.OH, p.OW = conv["IH"], conv["IW"], conv["OH"], conv["OW"]
        p.stride, p.pad, p.ups = conv.get("stride", 1), conv.get("pad", 1), conv.get("ups", 0)
        if pw.taps != 9 or a.shape[1] < pw.Cin:
            raise ValueError("conv3x3 weight / activation mismatch")
    elif tconv is not None:
        p.mode = 2
        p.T, p.HW = tconv["T"], tconv["HW"]
        if pw.taps != 3:
            raise ValueError("tconv weight mismatch")
    else:
        p.mode = 0
        if a.shape[1] < pw.K:
            raise ValueError(f"gemm: activation has {a.shape[1]} columns, weight K={pw.K}")
    n_out = pw.N // 2 if geglu else pw.N
    if out.shape[1] < n_out:
        raise ValueError(f"gemm: out has {out.shape[1]} columns, need {n_out}")
    # row extents the launch addresses (the C ABI takes raw pointers)
    if out.shape[0] < p.M or (residual is not None and (residual.shape[0] < p.M or residual.shape[1] < n_out)):
        raise ValueError(f"gemm: out/residual have fewer than M={p.M} rows (or residual fewer than {n_out} columns)")
    if p.mode == 1:
        if p.M % (p.OH * p.OW) != 0 or a.shape[0] < (p.M // (p.OH * p.OW)) * p.IH * p.IW:
            raise ValueError(f"conv3x3: M={p.M} output rows need {p.M // (p.OH * p.OW)} frames of {p.IH}x{p.IW} input rows, "
                             f"activation has {a.shape[0]}")
    elif a.shape[0] < p.M:
        raise ValueError(f"gemm: activation has {a.shape[0]} rows, M={p.M}")
    if p.mode == 2 and (p.M % (p.T * p.HW) != 0 or a.shape[1] < pw.Cin):
        raise ValueError("tconv: M must be a multiple of T*HW and the activation at least Cin wide")
    if rowvec is not None and (rowvec.shape[0] * rows_per_vec < p.M or rowvec.shape[1] < n_out):
        raise ValueError("gemm: rowvec does not cover every row group / output column")
    if _TRACE is not None:
        variant = "dc_gemm_conv"
        k_real = pw.k_real if pw.k_real else pw.K
        flops = 2.0 * p.M * pw.N * k_real
        esz = 4 if out_f32 else 2
        nbytes = 2.0 * p.M * (k_real if p.mode == 0 else pw.Cin) + 2.0 * pw.N * pw.K + esz * p.M * n_out \
            + (2.0 * p.M * n_out if residual is not None else 0)
        _launch(variant, flops, nbytes, _hip.lib().dc_gemm_conv, C.byref(p), stream_ptr(),
                tag=(p.mode, p.M, pw.N, k_real))
    else:
        check(_hip.lib().dc_gemm_conv(C.byref(p), stream_ptr()), "dc_gemm_conv")
    return out


_gemm_ws = {}


def _gemm_workspace(device, stream):
    """One scratch buffer per (device, stream) for split-K partial sums: launches on one stream use it one after
    another; a graph replay on its private stream and eager work on another stream never share partials."""
    key = (device.index, stream)
    buf = _gemm_ws.get(key)
    if buf is None:
        buf = torch.empty(int(_hip.lib().dc_gemm_workspace_bytes()), dtype=torch.uint8, device=device)
        _gemm_ws[key] = buf
    return buf


class Arena:
    """Named, shape-keyed device scratch: allocated on first use, stable afterwards (graph-safe).

    DC_ARENA_GUARD=1 (debugging / tests): every buffer is carved out of a larger allocation with `GUARD` sentinel
    rows before and after it; `check()` raises if a launch wrote outside its buffer. The kernels take raw pointers,
    so this is the only place an overrun of a scratch buffer can be made visible."""
    GUARD = 8

    def __init__(self):
        self._bufs = {}
        self._guarded = {}
        self.guard = os.environ.get("DC_ARENA_GUARD", "0") == "1"

    def get(self, tag, rows, cols, dtype=_BF16, device=None, zero=False):
        key = (tag, rows, cols, dtype)
        b = self._bufs.get(key)
        if b is None:
            if self.guard:
                G = self.GUARD
                whole = torch.empty((rows + 2 * G, cols), dtype=dtype, device=device)
                whole.fill_(self._sentinel(dtype))
                b = whole[G:G + rows]
                if zero:
                    b.zero_()
                self._guarded[key] = whole
            else:
                b = (torch.zeros if zero else torch.empty)((rows, cols), dtype=dtype, device=device)
            self._bufs[key] = b
        return b

    @staticmethod
    def _sentinel(dtype):
        return 12345.0 if dtype.is_floating_point else 0x5a

    def check(self):
        """Verify the guard rows of every buffer (guard mode only). Synchronises."""
        bad = []
        for (tag, rows, cols, dtype), whole in self._guarded.items():
            G = self.GUARD
            ref = torch.full((1,), self._sentinel(dtype), dtype=dtype, device=whole.device)
            if not (bool((whole[:G] == ref).all()) and bool((whole[G + rows:] == ref).all())):
                bad.append(f"{tag}[{rows}x{cols} {dtype}]")
        if bad:
            raise RuntimeError("scratch buffers overrun: " + ", ".join(bad))
        return len(self._guarded)

    def nbytes(self):
        return sum(b.numel() * b.element_size() for b in self._bufs.values())


_gn_ws = {}


_gn_ws_retired = []


def _gn_workspace(device, nbytes):
    """GroupNorm statistics scratch per (device, stream). A buffer that was ever handed out is never released: a captured
    hipGraph keeps its address, so a larger request allocates a new buffer and the old one is parked, not freed."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _gn_ws.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        if buf is not None:
            _gn_ws_retired.append(buf)
        buf = torch.empty((max(nbytes, 4 << 20) + 3) // 4, dtype=torch.float32, device=device)
        _gn_ws[key] = buf
    return buf


def _need(t, n, name):
    """The C ABI takes raw pointers: refuse operands smaller than the extent the kernel will touch."""
    if t is not None and t.numel() < n:
        raise ValueError(f"{name}: {t.numel()} elements, the launch reads/writes {n}")


def _need_rows(t, rows, cols, name):
    if t.shape[0] < rows or t.shape[1] < cols:
        raise ValueError(f"{name}: shape {tuple(t.shape)}, the launch addresses [{rows}, {cols}]")


def groupnorm(x, y, gamma, beta, *, groups, n_inst, rows_per_inst, eps, silu):
    _rows(x, "x"); _rows(y, "y")
    _need_rows(x, n_inst * rows_per_inst, gamma.numel(), "x"); _need_rows(y, n_inst * rows_per_inst, gamma.numel(), "y")
    _need(beta, gamma.numel(), "beta")
    Cc = gamma.numel()
    l = _hip.lib()
    ws = _gn_workspace(x.device, int(l.dc_groupnorm_workspace_bytes(n_inst, groups, rows_per_inst)))
    _launch("groupnorm(2-3 kernels)", 0.0, 6.0 * n_inst * rows_per_inst * Cc, l.dc_groupnorm, _ptr(x), x.stride(0), _ptr(y),
            y.stride(0), _ptr(gamma), _ptr(beta), Cc, groups, n_inst, rows_per_inst, eps, 1 if silu else 0, _ptr(ws),
            stream_ptr())
    return y


def layernorm(x, y, gamma, beta, eps=1e-5):
    _rows(x, "x"); _rows(y, "y")
    _need_rows(x, x.shape[0], gamma.numel(), "x"); _need_rows(y, x.shape[0], gamma.numel(), "y"); _need(beta, gamma.numel(), "beta")
    _launch("layernorm", 0.0, 4.0 * x.shape[0] * gamma.numel(), _hip.lib().dc_layernorm, _ptr(x), x.stride(0), _ptr(y),
            y.stride(0), _ptr(gamma), _ptr(beta), x.shape[0], gamma.numel(), eps, stream_ptr())
    return y


def flash_attn(q, k, v, o, *, batch, heads, Lq, Lk, scale, accumulate=False, acc_scale=1.0, q_bstride=None,
               kv_bstride=None):
    """q/o rows [batch*Lq, >=heads*64]; k/v rows [batch*Lk, ...] (views into a fused qkv buffer are fine).
    q_bstride / kv_bstride: rows between consecutive batch items (default Lq / Lk)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        _rows(t, n)
    qb = Lq if q_bstride is None else q_bstride
    kb = Lk if kv_bstride is None else kv_bstride
    _need_rows(q, (batch - 1) * qb + Lq, heads * 64, "q"); _need_rows(o, (batch - 1) * qb + Lq, heads * 64, "o")
    _need_rows(k, (batch - 1) * kb + Lk, heads * 64, "k"); _need_rows(v, (batch - 1) * kb + Lk, heads * 64, "v")
    _launch("flash_attn_d64(self)" if Lk > 128 else "flash_attn_d64(cross)", 4.0 * batch * heads * Lq * Lk * 64,
            2.0 * batch * heads * 64 * (2 * Lq + 2 * Lk), _hip.lib().dc_flash_attn_d64, _ptr(q), _ptr(k), _ptr(v), _ptr(o),
            q.stride(0), k.stride(0), v.stride(0), o.stride(0), batch, heads, Lq, Lk,
            Lq if q_bstride is None else q_bstride, Lk if kv_bstride is None else kv_bstride, scale,
            1 if accumulate else 0, acc_scale, stream_ptr())
    return o


def cross_attn_dual(q, k, v, k2, v2, o, *, batch, heads, Lq, Lk, Lk2, scale, scale2, kv_bstride):
    """o = attn(q; k, v) + scale2 * attn(q; k2, v2): text + image cross-attention of one query tensor in one launch.
    k/v/k2/v2 are column views of one projection buffer (same row stride), kv_bstride rows per batch item."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (k2, "k2"), (v2, "v2"), (o, "o")):
        _rows(t, n)
    if len({k.stride(0), v.stride(0), k2.stride(0), v2.stride(0)}) != 1:
        raise ValueError("cross_attn_dual: k, v, k2, v2 must share one row stride")
    _need_rows(q, batch * Lq, heads * 64, "q"); _need_rows(o, batch * Lq, heads * 64, "o")
    _need_rows(k, (batch - 1) * kv_bstride + Lk, heads * 64, "k"); _need_rows(v, (batch - 1) * kv_bstride + Lk, heads * 64, "v")
    _need_rows(k2, (batch - 1) * kv_bstride + Lk2, heads * 64, "k2"); _need_rows(v2, (batch - 1) * kv_bstride + Lk2, heads * 64, "v2")
    _launch("flash_attn_d64(cross)", 4.0 * batch * heads * Lq * (Lk + Lk2) * 64, 2.0 * batch * heads * 64 * (2 * Lq + 2 * (Lk + Lk2)),
            _hip.lib().dc_cross_attn_dual_d64, _ptr(q), _ptr(k), _ptr(v), _ptr(k2), _ptr(v2), _ptr(o), q.stride(0), k.stride(0),
            o.stride(0), batch, heads, Lq, Lk, Lk2, Lq, kv_bstride, scale, scale2, stream_ptr())
    return o


def temporal_attn(qkv, o, *, B, T, HW, heads, scale):
    _rows(qkv, "qkv"); _rows(o, "o")
    _need_rows(qkv, B * T * HW, 3 * heads * 64, "qkv"); _need_rows(o, B * T * HW, heads * 64, "o")
    _launch("temporal_attn_d64", 4.0 * B * HW * heads * T * T * 64, 2.0 * B * T * HW * heads * 64 * 4,
            _hip.lib().dc_temporal_attn_d64, _ptr(qkv), qkv.stride(0), _ptr(o), o.stride(0), B, T, HW, heads, scale,
            stream_ptr())
    return o


def ff2_permuted(weight, device):
    """ff.net.2.weight [320, 1280] -> bf16 [384, 1280] in the k order dc_ff_geglu_fused320 reads: inside every
    32-channel chunk, position 16 s + 8 h + e holds channel 8 (2 s + e // 4) + 4 h + e % 4."""
    w = weight.detach()
    N, K = w.shape
    pos = torch.arange(32)
    s_, h_, e_ = pos // 16, (pos // 8) % 2, pos % 8
    chan = 8 * (2 * s_ + e_ // 4) + 4 * h_ + e_ % 4
    idx = (torch.arange(K // 32)[:, None] * 32 + chan[None, :]).reshape(-1)
    n_pad = (N + 127) // 128 * 128
    out = torch.zeros((n_pad, K), dtype=_BF16, device=device)
    out[:N] = w[:, idx].to(device=device, dtype=_BF16)
    return out


def ff_geglu_fused320(x, pw1, w2p, b2, out, residual=None, ln=None, ln_eps=1e-5):
    """out = FeedForward_GEGLU(LayerNorm(x) if ln else x) (+ residual) for dim 320 in one launch; pw1 =
    PackedWeight.linear(ff.net.0.proj); ln = (gamma, beta) fp32 or None."""
    _rows(x, "x"); _rows(out, "out")
    M = x.shape[0]
    if pw1.K != 320 or pw1.N != 2560 or pw1.bias is None or tuple(w2p.shape[1:]) != (1280,) or w2p.shape[0] < 320:
        raise ValueError("ff_geglu_fused320: dim must be 320 (ff1 [2560, 320] with bias, ff2 [320, 1280])")
    _need_rows(x, M, 320, "x"); _need_rows(out, M, 320, "out")
    if residual is not None:
        _rows(residual, "residual"); _need_rows(residual, M, 320, "residual")
    if ln is not None:
        _need(ln[0], 320, "ln gamma"); _need(ln[1], 320, "ln beta")
    flops = 2.0 * M * (2560 + 1280) * 320
    nbytes = 2.0 * M * 320 * (3 if residual is not None else 2) + 2.0 * (2560 * 320 + 320 * 1280)
    _launch("ff_geglu_fused320", flops, nbytes, _hip.lib().dc_ff_geglu_fused320, _ptr(x), x.stride(0),
            _ptr(None if ln is None else ln[0]), _ptr(None if ln is None else ln[1]), ln_eps, _ptr(pw1.w), _ptr(pw1.bias),
            _ptr(w2p), _ptr(b2), _ptr(residual), 0 if residual is None else residual.stride(0), _ptr(out), out.stride(0), M,
            stream_ptr())
    return out


def ff_geglu_proj_fused320(x, pw1, w2p, b2, wp, bp, residual2, out, ln=None, ln_eps=1e-5):
    """out = residual2 + Linear_p(x + FeedForward_GEGLU(LayerNorm(x) if ln else x)) for dim 320 in one launch (the last two
    steps of a transformer: FeedForward with its residual, proj_out with the transformer's residual); wp =
    ff2_permuted(proj_out.weight)."""
    _rows(x, "x"); _rows(out, "out"); _rows(residual2, "residual2")
    M = x.shape[0]
    if pw1.K != 320 or pw1.N != 2560 or pw1.bias is None or tuple(w2p.shape[1:]) != (1280,) or w2p.shape[0] < 320 \
            or tuple(wp.shape[1:]) != (320,) or wp.shape[0] < 320:
        raise ValueError("ff_geglu_proj_fused320: dim must be 320 (ff1 [2560, 320] with bias, ff2 [320, 1280], proj [320, 320])")
    if x.data_ptr() == out.data_ptr():
        raise ValueError("ff_geglu_proj_fused320: out must not alias x")
    _need_rows(x, M, 320, "x"); _need_rows(out, M, 320, "out"); _need_rows(residual2, M, 320, "residual2")
    _need(b2, 320, "b2"); _need(bp, 320, "bp")
    if ln is not None:
        _need(ln[0], 320, "ln gamma"); _need(ln[1], 320, "ln beta")
    _launch("ff_geglu_proj_fused320", 2.0 * M * (2560 + 1280 + 320) * 320, 2.0 * M * 320 * 3 + 2.0 * (2560 * 320 + 320 * 1280 + 320 * 320),
            _hip.lib().dc_ff_geglu_proj_fused320, _ptr(x), x.stride(0), _ptr(None if ln is None else ln[0]),
            _ptr(None if ln is None else ln[1]), ln_eps, _ptr(pw1.w), _ptr(pw1.bias), _ptr(w2p), _ptr(b2), _ptr(wp), _ptr(bp),
            _ptr(residual2), residual2.stride(0), _ptr(out), out.stride(0), M, stream_ptr())
    return out


def ln_linear(x, pw, out, ln=None, ln_eps=1e-5):
    """out = Linear(LayerNorm(x) if ln else x) for dim 320 / 640 in one launch; pw = PackedWeight.linear (N % 32 == 0);
    ln = (gamma, beta) fp32 or None."""
    _rows(x, "x"); _rows(out, "out")
    M, K = x.shape[0], pw.K
    if K not in (320, 640) or pw.N % 32:
        raise ValueError("ln_linear: K must be 320 or 640 and N a multiple of 32")
    _need_rows(x, M, K, "x"); _need_rows(out, M, pw.N, "out")
    if ln is not None:
        _need(ln[0], K, "ln gamma"); _need(ln[1], K, "ln beta")
    _launch("ln_linear", 2.0 * M * pw.N * K, 2.0 * M * (K + pw.N) + 2.0 * pw.N * K, _hip.lib().dc_ln_linear,
            _ptr(x), x.stride(0), K, _ptr(None if ln is None else ln[0]), _ptr(None if ln is None else ln[1]), ln_eps,
            _ptr(pw.w), _ptr(pw.bias), _ptr(out), out.stride(0), M, pw.N, stream_ptr())
    return out


def linear_residual(x, pw, residual, out):
    """out = residual + Linear(x) for dim K = 320 / 640 in the X-stationary kernel (pw = PackedWeight.linear, N % 32 == 0);
    residual may be `out` itself."""
    _rows(x, "x"); _rows(out, "out"); _rows(residual, "residual")
    M, K = x.shape[0], pw.K
    if K not in (320, 640) or pw.N % 32:
        raise ValueError("linear_residual: K must be 320 or 640 and N a multiple of 32")
    _need_rows(x, M, K, "x"); _need_rows(out, M, pw.N, "out"); _need_rows(residual, M, pw.N, "residual")
    _launch("linear_residual", 2.0 * M * pw.N * K, 2.0 * M * (K + 2 * pw.N) + 2.0 * pw.N * K, _hip.lib().dc_linear_residual,
            _ptr(x), x.stride(0), K, _ptr(pw.w), _ptr(pw.bias), _ptr(residual), residual.stride(0), _ptr(out), out.stride(0),
            M, pw.N, stream_ptr())
    return out


def ln_qkv_temporal_attn(x, ln, pw_qkv, out, *, B, T, HW, scale, ln_eps=1e-5):
    """out = temporal self-attention (over the T = 16 frames of a position) of LayerNorm(x), q/k/v projected in the same
    launch; dim 320 = 5 heads x 64 or 640 = 10 heads x 64; rows ordered (clip, frame, position)."""
    _rows(x, "x"); _rows(out, "out")
    M = B * T * HW
    Cc = pw_qkv.K
    if Cc not in (320, 640) or pw_qkv.N != 3 * Cc or pw_qkv.bias is not None or T != 16 or HW % 8:
        raise ValueError("ln_qkv_temporal_attn: dim 320 / 640 (qkv weight [3 C, C], no bias), T = 16, HW % 8 == 0")
    if x.data_ptr() == out.data_ptr():
        raise ValueError("ln_qkv_temporal_attn: out must not alias x")
    _need_rows(x, M, Cc, "x"); _need_rows(out, M, Cc, "out"); _need(ln[0], Cc, "ln gamma"); _need(ln[1], Cc, "ln beta")
    fn = _hip.lib().dc_ln_qkv_temporal_attn320 if Cc == 320 else _hip.lib().dc_ln_qkv_temporal_attn640
    _launch(f"ln_qkv_temporal_attn{Cc}", 2.0 * M * 3 * Cc * Cc + 4.0 * M * T * Cc, 4.0 * M * Cc + 2.0 * 3 * Cc * Cc,
            fn, _ptr(x), x.stride(0), _ptr(ln[0]), _ptr(ln[1]), ln_eps, _ptr(pw_qkv.w),
            _ptr(out), out.stride(0), B, T, HW, scale, stream_ptr())
    return out


def ln_qkv_temporal_attn320(x, ln, pw_qkv, out, *, B, T, HW, scale, ln_eps=1e-5):
    if pw_qkv.K != 320:
        raise ValueError("ln_qkv_temporal_attn320: dim 320 (qkv weight [960, 320], no bias), T = 16, HW % 8 == 0")
    return ln_qkv_temporal_attn(x, ln, pw_qkv, out, B=B, T=T, HW=HW, scale=scale, ln_eps=ln_eps)


def gn_silu_tconv3(x, gamma, beta, stats, pw, out, *, B, T, HW, groups=32, residual=None):
    """out = tconv3(silu(GroupNorm(x))) (+ residual) for 320 / 640 input channels, statistics from groupnorm_stats over the
    clips (n_inst = B, rows_per_inst = T * HW); pw = PackedWeight.tconv3; rows ordered (clip, frame, position)."""
    _rows(x, "x"); _rows(out, "out")
    M = B * T * HW
    C = pw.w.shape[1] // 3
    if C not in (320, 640) or pw.w.shape[1] != 3 * C or pw.N % 32 or pw.bias is None or T != 16 or HW % 8:
        raise ValueError("gn_silu_tconv3: 320 / 640 input channels (weight [N, 3 C] with bias), N % 32 == 0, T = 16, HW % 8 == 0")
    if x.data_ptr() == out.data_ptr():
        raise ValueError("gn_silu_tconv3: out must not alias x")
    _need_rows(x, M, C, "x"); _need_rows(out, M, pw.N, "out")
    _need(gamma, C, "gamma"); _need(beta, C, "beta"); _need(stats, B * groups * 2, "stats")
    if residual is not None:
        _rows(residual, "residual"); _need_rows(residual, M, pw.N, "residual")
    _launch("gn_silu_tconv3", 2.0 * M * pw.N * 3 * C, 2.0 * M * (C + pw.N * (2 if residual is not None else 1)) + 2.0 * pw.N * 3 * C,
            _hip.lib().dc_gn_silu_tconv3, _ptr(x), x.stride(0), C, _ptr(gamma), _ptr(beta), _ptr(stats), groups, _ptr(pw.w),
            _ptr(pw.bias), _ptr(residual), 0 if residual is None else residual.stride(0), _ptr(out), out.stride(0), B, T, HW,
            pw.N, stream_ptr())
    return out


def groupnorm_stats(x, stats, *, groups, n_inst, rows_per_inst, eps):
    """(mean, rstd) per (instance, group) of channels-last rows -> stats fp32 [n_inst, groups, 2] (for gn_linear320)."""
    _rows(x, "x")
    Cc = x.shape[1]
    _need_rows(x, n_inst * rows_per_inst, Cc, "x"); _need(stats, n_inst * groups * 2, "stats")
    l = _hip.lib()
    ws = _gn_workspace(x.device, int(l.dc_groupnorm_workspace_bytes(n_inst, groups, rows_per_inst)))
    _launch("groupnorm_stats(2 kernels)", 0.0, 2.0 * n_inst * rows_per_inst * Cc, l.dc_groupnorm_stats, _ptr(x), x.stride(0),
            Cc, groups, n_inst, rows_per_inst, eps, _ptr(ws), _ptr(stats), stream_ptr())
    return stats


def gn_linear(x, gamma, beta, stats, pw, out, *, groups, rows_per_inst):
    """out = Linear(GroupNorm(x)) for dim 320 / 640 with the statistics of groupnorm_stats, normalisation applied in
    registers."""
    _rows(x, "x"); _rows(out, "out")
    M, K = x.shape[0], pw.K
    if K not in (320, 640) or pw.N % 32 or rows_per_inst % 128 or M % rows_per_inst:
        raise ValueError("gn_linear: K in (320, 640), N % 32 == 0, rows_per_inst % 128 == 0, M % rows_per_inst == 0")
    _need_rows(x, M, K, "x"); _need_rows(out, M, pw.N, "out")
    _need(gamma, K, "gamma"); _need(beta, K, "beta"); _need(stats, (M // rows_per_inst) * groups * 2, "stats")
    _launch("gn_linear", 2.0 * M * pw.N * K, 2.0 * M * (K + pw.N) + 2.0 * pw.N * K, _hip.lib().dc_gn_linear,
            _ptr(x), x.stride(0), K, _ptr(gamma), _ptr(beta), _ptr(stats), groups, rows_per_inst, _ptr(pw.w), _ptr(pw.bias),
            _ptr(out), out.stride(0), M, pw.N, stream_ptr())
    return out


def attn_small(q, k, v, o, *, batch, heads, Lq, Lk, d, scale, causal=False):
    """Any-head-width attention (CLIP towers): q/o rows [batch*Lq, >= heads*d], k/v rows [batch*Lk, >= heads*d]."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (o, "o")):
        _rows(t, n)
    _need_rows(q, batch * Lq, heads * d, "q"); _need_rows(o, batch * Lq, heads * d, "o")
    _need_rows(k, batch * Lk, heads * d, "k"); _need_rows(v, batch * Lk, heads * d, "v")
    _launch("attn_small", 4.0 * batch * heads * Lq * Lk * d, 2.0 * batch * heads * d * (2 * Lq + 2 * Lk),
            _hip.lib().dc_attn_small, _ptr(q), _ptr(k), _ptr(v), _ptr(o), q.stride(0), k.stride(0), v.stride(0), o.stride(0),
            batch, heads, Lq, Lk, d, scale, 1 if causal else 0, stream_ptr())
    return o


def clip_preprocess(img, out_hw=(224, 224), antialias=True, mean=(0.48145466, 0.4578275, 0.40821073),
                    std=(0.26862954, 0.26130258, 0.27577711)):
    """img fp32 [N,3,H,W] in [-1,1] -> fp32 [N,3,OH,OW] (bicubic align_corners resize w/ antialias blur, CLIP mean/std)."""
    img = img.to(torch.float32).contiguous()
    N, Cc, H, W = img.shape
    out = torch.empty((N, Cc, out_hw[0], out_hw[1]), dtype=torch.float32, device=img.device)
    blur = antialias and max(H / out_hw[0], W / out_hw[1]) > 1.0
    t0 = torch.empty_like(img) if blur else None
    t1 = torch.empty_like(img) if blur else None
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(_hip.lib().dc_clip_preprocess(_ptr(img), _ptr(t0), _ptr(t1), _ptr(out), N, Cc, H, W, out_hw[0], out_hw[1],
                                        1 if antialias else 0, m3, s3, stream_ptr()), "dc_clip_preprocess")
    return out


def patchify(img, rows, *, patch):
    N, Cc, H, W = img.shape
    _need_rows(rows, N * (H // patch) * (W // patch), Cc * patch * patch, "rows")
    check(_hip.lib().dc_patchify(_ptr(img), _ptr(rows), N, Cc, H, W, patch, rows.stride(0), stream_ptr()), "dc_patchify")
    return rows


def embed_tokens(tokens, table, pos, out):
    B, L = tokens.shape
    D = table.shape[1]
    _need_rows(out, B * L, D, "out"); _need_rows(pos, L, D, "pos")
    if out.stride(0) != D or table.stride(0) != D or pos.stride(0) != D:
        raise ValueError("embed_tokens: dense rows expected")
    check(_hip.lib().dc_embed_tokens(_ptr(tokens), _ptr(table), _ptr(pos), _ptr(out), B, L, D, table.shape[0], stream_ptr()),
          "dc_embed_tokens")
    return out


def gemv_small(x, pw, out, *, act_in=0, act_out=0, accumulate=False):
    """x [M<=8, K] fp32, out [M, N] fp32."""
    if x.shape[0] > 8 or x.shape[1] < pw.K or out.shape[0] < x.shape[0] or out.shape[1] < pw.N:
        raise ValueError(f"gemv_small: x {tuple(x.shape)}, weight [{pw.N}, {pw.K}], out {tuple(out.shape)}")
    check(_hip.lib().dc_gemv_small(_ptr(x), x.stride(0), _ptr(pw.w), _ptr(pw.bias), _ptr(out), out.stride(0),
                                   x.shape[0], pw.N, pw.K, act_in, act_out, 1 if accumulate else 0, stream_ptr()),
          "dc_gemv_small")
    return out


def timestep_embedding(t_table, out, dim, *, t_index=None, t_stride=0, max_period=10000.0):
    if out.shape[1] < dim or t_table.numel() < out.shape[0]:
        raise ValueError(f"timestep_embedding: out {tuple(out.shape)}, dim {dim}, table of {t_table.numel()} entries")
    check(_hip.lib().dc_timestep_embedding(_ptr(t_table), _ptr(t_index), t_stride, _ptr(out), out.shape[0], dim,
                                           max_period, stream_ptr()), "dc_timestep_embedding")
    return out


def pack_latent(x, cc, out, *, B, Cx, Cc, T, HW, nrep=1):
    _need(x, B * Cx * T * HW, "x"); _need(cc, B * Cc * T * HW, "c_concat"); _need_rows(out, nrep * B * T * HW, Cx + Cc, "out")
    check(_hip.lib().dc_pack_latent(_ptr(x), _ptr(cc), _ptr(out), B, Cx, Cc, T, HW, out.stride(0), nrep,
                                    stream_ptr()), "dc_pack_latent")
    return out


def nchw_to_rows(x, out, *, N, Cc, HW, scale=1.0):
    _need(x, N * Cc * HW, "x"); _need_rows(out, N * HW, Cc, "out")
    check(_hip.lib().dc_nchw_to_rows(_ptr(x), _ptr(out), N, Cc, HW, out.stride(0), scale, stream_ptr()),
          "dc_nchw_to_rows")
    return out


def rows_to_nchw(rows, y, *, N, Cc, HW, scale=1.0):
    _need_rows(rows, N * HW, Cc, "rows"); _need(y, N * Cc * HW, "y")
    check(_hip.lib().dc_rows_to_nchw(_ptr(rows), rows.stride(0), 1 if rows.dtype == torch.float32 else 0, _ptr(y),
                                     N, Cc, HW, scale, stream_ptr()), "dc_rows_to_nchw")
    return y


def im2col3x3_c8(x, out, *, n_img, H, W):
    """rows [n_img*H*W, >= 8] (first 8 channels) -> rows [n_img*H*W, 128]: 9 taps x 8 channels + zeros (dc_im2col3x3_c8)."""
    _rows(x, "x"); _rows(out, "out")
    M = n_img * H * W
    _need_rows(x, M, 8, "x"); _need_rows(out, M, 128, "out")
    _launch("im2col3x3_c8", 0.0, 2.0 * M * (16 + 256), _hip.lib().dc_im2col3x3_c8, _ptr(x), x.stride(0), _ptr(out), out.stride(0),
            n_img, H, W, stream_ptr())
    return out


def copy2d(src, dst, cols=None):
    _rows(src, "src"); _rows(dst, "dst")
    cols = src.shape[1] if cols is None else cols
    _launch("copy2d", 0.0, 4.0 * src.shape[0] * cols, _hip.lib().dc_copy2d, _ptr(src), src.stride(0), _ptr(dst),
            dst.stride(0), src.shape[0], cols, stream_ptr())
    return dst


def transpose(src, dst, rows=None, cols=None):
    """dst[c, r] = src[r, c] (bf16)."""
    rows = src.shape[0] if rows is None else rows
    cols = src.shape[1] if cols is None else cols
    check(_hip.lib().dc_transpose(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), rows, cols, stream_ptr()),
          "dc_transpose")
    return dst


def add_rows(a, b, y):
    check(_hip.lib().dc_add_rows(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(y), y.stride(0), a.shape[0],
                                 a.shape[1], stream_ptr()), "dc_add_rows")
    return y


def build_context(ctx, out, *, B, T, n_text, L, D):
    _need(ctx, B * (n_text + T * L) * D, "context"); _need(out, B * T * (n_text + L) * D, "out")
    check(_hip.lib().dc_build_context(_ptr(ctx), _ptr(out), B, T, n_text, L, D, stream_ptr()), "dc_build_context")
    return out


def softmax_rows(x, y):
    check(_hip.lib().dc_softmax_rows(_ptr(x), x.stride(0), _ptr(y), y.stride(0), x.shape[0], x.shape[1],
                                     stream_ptr()), "dc_softmax_rows")
    return y


def vae_sample(moments, noise, z, *, N, zc, HW, scale):
    _need_rows(moments, N * HW, 2 * zc, "moments"); _need(noise, N * zc * HW, "noise"); _need(z, N * zc * HW, "z")
    check(_hip.lib().dc_vae_sample(_ptr(moments), moments.stride(0), _ptr(noise), _ptr(z), N, zc, HW, scale,
                                   stream_ptr()), "dc_vae_sample")
    return z


def ddim_step(tables, e_cond, e_uncond, e_img, x, noise, x_prev, pred_x0, workspace, *, B, Cc, THW, index=0,
              step_index=None, v_param=False, cfg_scale=1.0, cfg_img=1.0, guidance_rescale=0.0, temperature=1.0,
              e_nchw=False, ld_e=None, noise_step_stride=0):
    """tables: dict of fp32 device vectors (a_t, a_prev, sigma_t, sqrt_one_minus_at[, sqrt_acp_t, sqrt_1macp_t,
    scale_ratio]) indexed by the DDIM index."""
    p = DcDdimParams()
    for k in ("a_t", "a_prev", "sigma_t", "sqrt_one_minus_at", "sqrt_acp_t", "sqrt_1macp_t", "scale_ratio"):
        t = tables.get(k)
        setattr(p, k, 0 if t is None else t.data_ptr())
    p.step_index = 0 if step_index is None else step_index.data_ptr()
    p.index, p.v_param = index, 1 if v_param else 0
    p.cfg_scale, p.cfg_img, p.guidance_rescale, p.temperature = cfg_scale, cfg_img, guidance_rescale, temperature
    p.e_nchw, p.noise_step_stride = 1 if e_nchw else 0, noise_step_stride
    if ld_e is None:
        ld_e = 0 if e_nchw else e_cond.stride(0)
    check(_hip.lib().dc_ddim_step(C.byref(p), _ptr(e_cond), _ptr(e_uncond), _ptr(e_img), ld_e, _ptr(x),
                                  _ptr(noise), _ptr(x_prev), _ptr(pred_x0), B, Cc, THW, _ptr(workspace),
                                  stream_ptr()), "dc_ddim_step")
    return x_prev, pred_x0


def mask_blend(img, x0, mask, qnoise, tables, *, index=0, step_index=None, clean=False, noise_step_stride=0):
    """img = orig*mask + (1-mask)*img in place, orig = x0 or its q_sample at the step's timestep (fp32, same shapes)."""
    n = img.numel()
    for t, nm in ((x0, "x0"), (mask, "mask")):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise ValueError(f"mask_blend: {nm} must be contiguous fp32 of the latent's shape")
    if img.dtype != torch.float32 or not img.is_contiguous():
        raise ValueError("mask_blend: latent must be contiguous fp32")
    if not clean:
        # with a device step counter the kernel reads qnoise[step * noise_step_stride + i], step < the number of table rows
        steps = int(tables["sqrt_acp_t"].numel()) if step_index is not None else 1
        _need(qnoise, (steps - 1) * noise_step_stride + n if step_index is not None else n, "qnoise")
    check(_hip.lib().dc_mask_blend(_ptr(img), _ptr(x0), _ptr(mask), _ptr(None if clean else qnoise),
                                   _ptr(tables.get("sqrt_acp_t")), _ptr(tables.get("sqrt_1macp_t")), _ptr(step_index), index,
                                   n, noise_step_stride, 1 if clean else 0, stream_ptr()), "dc_mask_blend")
    return img


def advance_counter(counter):
    check(_hip.lib().dc_advance_counter(_ptr(counter), stream_ptr()), "dc_advance_counter")


class DeviceGraph:
    """hipGraph capture/replay of a block of dc_* calls issued on a private stream (runtime.hip)."""

    def __init__(self):
        l = _hip.lib()
        s = C.c_void_p()
        check(l.dc_stream_create(C.byref(s)), "dc_stream_create")
        self._stream = s
        self.torch_stream = torch.cuda.ExternalStream(s.value)
        self._exec = None

    def capture(self, fn):
        l = _hip.lib()
        torch.cuda.synchronize()
        # the per-stream workspaces of the capture stream must exist before capture begins (no allocation inside a
        # capture): size them like the ones the eager warm-up used on the current stream
        cur = torch.cuda.current_stream()
        dev = cur.device
        cap = self.torch_stream.cuda_stream
        if (dev.index, cur.cuda_stream) in _gemm_ws:
            _gemm_workspace(dev, cap)
        like = _gn_ws.get((dev.index, cur.cuda_stream))
        if like is not None:
            with torch.cuda.stream(self.torch_stream):
                _gn_workspace(dev, like.numel() * 4)
        with torch.cuda.stream(self.torch_stream):
            check(l.dc_graph_begin_capture(self._stream), "dc_graph_begin_capture")
            try:
                fn()
            finally:
                ex = C.c_void_p()
                code = l.dc_graph_end_capture(self._stream, C.byref(ex))
            check(code, "dc_graph_end_capture")
        self._exec = ex
        return self

    def launch(self):
        check(_hip.lib().dc_graph_launch(self._exec, self._stream), "dc_graph_launch")

    def sync(self):
        check(_hip.lib().dc_stream_sync(self._stream), "dc_stream_sync")

    def __del__(self):
        try:
            l = _hip.lib()
            if self._exec is not None:
                l.dc_graph_destroy(self._exec)
            l.dc_stream_destroy(self._stream)
        except Exception:
            pass
