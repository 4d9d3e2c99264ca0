"""ctypes binding of libdcrafter_hip.so (C ABI declared in include/dcrafter_hip.h).

The product path has no fallback: if the shared library is missing or a symbol cannot be resolved, importing
this module's `lib()` raises. PyTorch is used only for device memory and the current-stream handle.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# DC_HIP_LIB: load an alternative build of the same library (instrumented tool builds, tools/gemm_stamps.py)
LIB_PATH = os.environ.get("DC_HIP_LIB") or os.path.join(_HERE, "csrc", "libdcrafter_hip.so")

DC_GEMM_OUT_F32 = 1
DC_GEMM_GEGLU = 2
DC_GEMM_GELU = 4


class DcGemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
        ("rowvec", C.c_void_p), ("residual", C.c_void_p),
        ("lda", C.c_int), ("ldc", C.c_int), ("ldr", C.c_int), ("rowvec_ld", C.c_int),
        ("rows_per_vec", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("n_pad", C.c_int),
        ("mode", C.c_int), ("Cin", C.c_int),
        ("IH", C.c_int), ("IW", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("ups", C.c_int),
        ("T", C.c_int), ("HW", C.c_int),
        ("flags", C.c_int), ("alpha", C.c_float),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong),
    ]


class DcDdimParams(C.Structure):
    _fields_ = [
        ("a_t", C.c_void_p), ("a_prev", C.c_void_p), ("sigma_t", C.c_void_p), ("sqrt_one_minus_at", C.c_void_p),
        ("sqrt_acp_t", C.c_void_p), ("sqrt_1macp_t", C.c_void_p), ("scale_ratio", C.c_void_p),
        ("step_index", C.c_void_p),
        ("index", C.c_int), ("v_param", C.c_int),
        ("cfg_scale", C.c_float), ("cfg_img", C.c_float), ("guidance_rescale", C.c_float),
        ("temperature", C.c_float),
        ("e_nchw", C.c_int), ("noise_step_stride", C.c_int64),
    ]


# name -> (restype, argtypes); must list every symbol include/dcrafter_hip.h declares
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
SIGNATURES = {
    "dc_gemm_conv": (_I, [C.POINTER(DcGemmParams), _P]),
    "dc_gemm_workspace_bytes": (_L, []),
    "dc_gemm_last_variant": (C.c_char_p, []),
    "dc_gemm_set_plan": (_I, [_I]),
    "dc_groupnorm": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P]),
    "dc_groupnorm_workspace_bytes": (_L, [_I, _I, _I]),
    "dc_layernorm": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _F, _P]),
    "dc_flash_attn_d64": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _F, _I, _F, _P]),
    "dc_ff_geglu_fused320": (_I, [_P, _I, _P, _P, _F, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "dc_ff_geglu_proj_fused320": (_I, [_P, _I, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "dc_ln_linear": (_I, [_P, _I, _I, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "dc_ln_qkv_temporal_attn320": (_I, [_P, _I, _P, _P, _F, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dc_ln_qkv_temporal_attn640": (_I, [_P, _I, _P, _P, _F, _P, _P, _I, _I, _I, _I, _F, _P]),
    "dc_gn_silu_tconv3": (_I, [_P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "dc_groupnorm_stats": (_I, [_P, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "dc_gn_linear": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "dc_flash_attn_set_mode": (_I, [_I, _F]),
    "dc_linear_residual": (_I, [_P, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "dc_cross_attn_dual_d64": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _F, _F, _P]),
    "dc_temporal_attn_d64": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "dc_gemv_small": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dc_timestep_embedding": (_I, [_P, _P, _I, _P, _I, _I, _F, _P]),
    "dc_pack_latent": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dc_nchw_to_rows": (_I, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "dc_rows_to_nchw": (_I, [_P, _I, _I, _P, _I, _I, _I, _F, _P]),
    "dc_im2col3x3_c8": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "dc_copy2d": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "dc_build_context": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dc_softmax_rows": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "dc_transpose": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "dc_add_rows": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "dc_vae_sample": (_I, [_P, _I, _P, _P, _I, _I, _I, _F, _P]),
    "dc_ddim_step": (_I, [C.POINTER(DcDdimParams), _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "dc_attn_small": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "dc_attn_small_lds_bytes": (_L, [_I, _I]),
    "dc_clip_preprocess": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P]),
    "dc_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dc_embed_tokens": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dc_frames_to_u8": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "dc_mask_blend": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _I, _P]),
    "dc_advance_counter": (_I, [_P, _P]),
    "dc_stream_create": (_I, [C.POINTER(_P)]),
    "dc_stream_destroy": (_I, [_P]),
    "dc_stream_sync": (_I, [_P]),
    "dc_graph_begin_capture": (_I, [_P]),
    "dc_graph_end_capture": (_I, [_P, C.POINTER(_P)]),
    "dc_graph_launch": (_I, [_P, _P]),
    "dc_graph_destroy": (_I, [_P]),
    "dc_event_create": (_I, [C.POINTER(_P)]),
    "dc_event_record": (_I, [_P, _P]),
    "dc_event_elapsed_ms": (_I, [_P, _P, C.POINTER(_F)]),
    "dc_event_destroy": (_I, [_P]),
    "dc_error_word_read": (_I, [C.POINTER(C.c_int), _I]),
    "dc_version": (C.c_char_p, []),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load (once) and return the C-ABI library. Raises if it is absent: there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} not found: build it with dynamicrafter_amd/csrc/build.sh "
                    "(or __graft_entry__.build()); the HIP extension is mandatory, there is no fallback")
            # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so): it must be the one in the process BEFORE this
            # library pulls in /opt/rocm's copy, or the kernels launch on a second runtime that has no device context
            # (hipErrorNoDevice at the first launch - seen when build() loaded the library ahead of `import torch`)
            import torch  # noqa: F401
            l = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(l, name)  # AttributeError if the symbol is missing
                fn.restype = res
                fn.argtypes = args
            _lib = l
    return _lib


ERROR_WORD_BITS = {1: "gemm_pipe320x16_kernel: LDS counter wait timed out",
                   2: "flash_attn_d64_pipe_kernel: K/V ring counter wait timed out",
                   4: "gemm_pp_kernel: LDS counter wait timed out"}


def check_error_word(what="", reset=True):
    """Read (and clear) the library's error word; raise if a kernel reported a timed-out counter wait since the last read -
    everything computed since then is suspect. Synchronises the device: call at the host's own sync points only."""
    w = C.c_int(0)
    check(lib().dc_error_word_read(C.byref(w), 1 if reset else 0), "dc_error_word_read")
    if w.value:
        names = [v for k, v in ERROR_WORD_BITS.items() if w.value & k] or [f"unknown bits {w.value:#x}"]
        raise RuntimeError(f"HIP kernels reported a failed wait ({what or 'error word'} = {w.value:#x}): " + "; ".join(names)
                           + " - results since the last check are NOT valid")


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}"
                           + (" (DC_ERR_SHAPE)" if code == -1 else " (DC_ERR_ARG)" if code == -2 else " (hipError_t)"))
