"""ctypes binding of libunsloth_amd.so (the C ABI declared in include/unsloth_amd.h).

Mirrors how the reference binds its only native dependency (bitsandbytes) at
unsloth/kernels/utils.py:198-202,242-284: function pointers off a shared library, raw device
pointers, C scalars, and the current stream of the tensor's device as an opaque pointer.

There is NO fallback: if the library is missing or a kernel rejects its arguments, the call
raises. A product path that silently ran PyTorch ops instead would void every parity claim.
"""
import ctypes
import os
from contextlib import nullcontext

import torch

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

HERE = os.path.dirname(os.path.abspath(__file__))
# UNSLOTH_AMD_LIB: another build of the SAME C ABI (A/B of a kernel file on one box: tools/gpu_r03_ae.sh); default in-tree
LIB_PATH = os.environ.get("UNSLOTH_AMD_LIB") or os.path.join(HERE, "lib", "libunsloth_amd.so")

UAMD_F32, UAMD_F16, UAMD_BF16 = 0, 1, 2
_DTYPE_CODE = {torch.float32: UAMD_F32, torch.float16: UAMD_F16, torch.bfloat16: UAMD_BF16}
_ERR = {-1: "unsupported dtype", -2: "invalid argument", -3: "pointer/stride not 16-byte aligned"}


class GemmGroup(ctypes.Structure):
    """uamd_gemm_group (include/unsloth_amd.h)."""

    _fields_ = [
        ("B", c_void_p), ("C", c_void_p), ("absmax", c_void_p), ("lora_xa", c_void_p),
        ("lora_b", c_void_p), ("ldb", c_int64), ("ldc", c_int64), ("ld_xa", c_int64),
        ("ld_lb", c_int64), ("N", c_int), ("R", c_int), ("lora_scale", c_float), ("_pad", c_int),
        ("lora_xk", c_void_p), ("lora_bk", c_void_p), ("ld_xk", c_int64), ("ld_bk", c_int64), ("Rk", c_int),
        ("_pad2", c_int), ("bias", c_void_p),
    ]


class GemvGroup(ctypes.Structure):
    """uamd_gemv_group (include/unsloth_amd.h)."""

    _fields_ = [
        ("W", c_void_p), ("absmax_u8", c_void_p), ("absmax_f32", c_void_p), ("code2", c_void_p), ("absmax2", c_void_p),
        ("y", c_void_p), ("lora_t", c_void_p), ("lora_b", c_void_p), ("bias", c_void_p),
        ("ldw", c_int64), ("ld_lb", c_int64), ("offset", c_float), ("lora_scale", c_float),
        ("N", c_int), ("R", c_int), ("blocksize2", c_int), ("lora_b_f32", c_int), ("y_f32", c_int), ("_pad", c_int),
    ]


class GemvPrologue(ctypes.Structure):
    """uamd_gemv_prologue (include/unsloth_amd.h)."""

    _fields_ = [("mode", c_int), ("Rt", c_int), ("w_f32", c_int), ("glu", c_int), ("x2", c_void_p), ("res", c_void_p),
                ("norm_w", c_void_p), ("h_out", c_void_p), ("a_rows", c_void_p), ("ld_a", c_int64), ("eps", c_float),
                ("t_off", c_int * 4), ("tag", ctypes.c_uint), ("sync", c_void_p), ("tag_dev", c_void_p)]


GEMV_SYNC_BYTES = 8 * 256               # UAMD_GEMV_SYNC_BYTES
TAG_STRIDE = 1024                       # UAMD_TAG_STRIDE


class LoraTnProblem(ctypes.Structure):
    """uamd_lora_tn_problem (include/unsloth_amd.h)."""

    _fields_ = [("P", c_void_p), ("Z", c_void_p), ("out", c_void_p), ("ldp", c_int64), ("ldz", c_int64),
                ("ldo", c_int64), ("N", c_int), ("R", c_int), ("out_nr", c_int), ("scale", c_float)]


# name -> (restype, argtypes); every symbol include/unsloth_amd.h declares
SIGNATURES = {
    "uamd_version": (c_int, []),
    "uamd_rms_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64,
                                       c_int64, c_float, c_int, c_int, c_int, c_void_p]),
    "uamd_rms_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                       c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "uamd_rms_layernorm_dw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64,
                                      c_int64, c_int, c_int, c_int, c_void_p]),
    "uamd_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64,
                                   c_int64, c_float, c_int, c_int, c_void_p]),
    "uamd_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64,
                                   c_int, c_int, c_void_p]),
    "uamd_rope_embedding": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "uamd_rope_embedding_qk": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                       c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "uamd_rope_embedding_qk_mrope": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                             c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int,
                                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "uamd_swiglu_fg": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_swiglu_DWf_DW_dfg": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_geglu_exact_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_geglu_exact_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_geglu_approx_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_geglu_approx_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_quick_gelu_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_quick_gelu_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_cross_entropy_forward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                           c_int, c_float, c_float, c_int, c_void_p]),
    "uamd_cross_entropy_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                            c_int64, c_int, c_float, c_float, c_int, c_void_p]),
    "cdequantize_blockwise_fp32": (None, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cdequantize_blockwise_bf16_nf4": (None, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cdequantize_blockwise_fp16_nf4": (None, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "cdequantize_blockwise_fp32_nf4": (None, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "uamd_dequantize_absmax": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int64,
                                       c_void_p]),
    "uamd_nf4_dequantize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                    c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int64,
                                    c_void_p]),
    "uamd_nf4_dequantize_multi": (c_int, [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                          ctypes.POINTER(c_int64), ctypes.POINTER(c_void_p), c_int, c_int, c_void_p]),
    "uamd_nf4_quantize": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "uamd_gemm_nt": (c_int, [c_void_p, c_int64, c_int, c_int, ctypes.POINTER(GemmGroup), c_int, c_int,
                             c_int, c_void_p]),
    "uamd_gemm_nt_256": (c_int, [c_void_p, c_int64, c_int, c_int, ctypes.POINTER(GemmGroup), c_int, c_int,
                                 c_int, c_void_p]),
    "uamd_gemm_nn_256": (c_int, [c_void_p, c_int64, c_int, c_int, ctypes.POINTER(GemmGroup), c_int, c_int,
                                 c_int, c_void_p]),
    "uamd_gemm_tn_256": (c_int, [c_void_p, c_int64, c_int, c_int, ctypes.POINTER(GemmGroup), c_int, c_int,
                                 c_int, c_void_p]),
    "uamd_set_tuning": (c_int, [c_int, c_int]),
    "uamd_gemm_nt_nf4": (c_int, [c_void_p, c_int64, c_int, c_int, ctypes.POINTER(GemmGroup), c_int,
                                 c_int, c_int, c_void_p]),
    "uamd_lora_xa": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                             c_int, c_int, c_int, c_void_p]),
    "uamd_lora_xa2": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                              c_int, c_int, c_int, c_void_p]),
    "uamd_lora_xa2k": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int,
                               c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "uamd_add_rms_layernorm_fwd": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_float,
                                                             c_int, c_int, c_void_p]),
    "uamd_add_rms_layernorm_bwd": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_int,
                                                             c_int, c_void_p]),
    "uamd_lora_prepare": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "uamd_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64] + [ctypes.c_double] * 8 + [c_int, c_void_p]),
    "uamd_glu_fwd_xa": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int64, c_int,
                                c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "uamd_glu_bwd_xa": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]
                        + [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int] * 2
                        + [c_int, c_void_p]),
    "uamd_glu_xa_workspace": (c_int64, [c_int, c_int, c_int, c_int]),
    "uamd_glu_fwd_xa_ws": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int64, c_int,
                                   c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "uamd_glu_bwd_xa_ws": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]
                           + [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int] * 2
                           + [c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "uamd_adamw_shard": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64] + [ctypes.c_double] * 8
                         + [c_int, c_void_p]),
    "uamd_lora_tn": (c_int, [ctypes.POINTER(LoraTnProblem), c_int, c_int, c_void_p, c_int64, c_int, c_void_p]),
    "uamd_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_int, c_int,
                              c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int, c_void_p]),
    "uamd_attn_fwd_band": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "uamd_attn_bwd": (c_int, [c_void_p] * 10 + [ctypes.POINTER(c_int64), c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_float, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "uamd_gemv": (c_int, [c_void_p, c_int, ctypes.POINTER(GemvGroup), c_int, c_int, c_int, c_int, c_void_p]),
    "uamd_gemv_fused": (c_int, [c_void_p, c_int, ctypes.POINTER(GemvGroup), c_int, c_int, c_int, c_int, c_void_p,
                                ctypes.POINTER(GemvPrologue)]),
    "uamd_rope_kv_append": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "uamd_attn_decode": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p,
                                 c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                 c_void_p]),
    "uamd_attn_decode_fused": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_float, ctypes.c_uint, c_void_p, c_int, c_void_p]),
    "uamd_argmax_f32": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "uamd_debug_mfma_probe": (c_int, [c_void_p, c_void_p]),
}

_LIB = None


def lib():
    """Load (once) and return the ctypes handle. Raises if the HIP library is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"unsloth_amd: {LIB_PATH} is missing. Build it with `python -m unsloth_amd._build` "
                "(hipcc, gfx950). There is no CPU / PyTorch fallback for the fused kernels."
            )
        # torch is already imported, so libamdhip64.so.7 resolves to the runtime torch uses.
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def dtype_code(dtype):
    try:
        return _DTYPE_CODE[dtype]
    except KeyError:
        raise TypeError(f"unsloth_amd kernels support fp32/fp16/bf16, got {dtype}") from None


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "unsloth_amd kernels run on MI355X (torch device 'cuda' under ROCm); got a "
                f"{t.device} tensor. The CPU oracle lives under oracle/ and is test-only."
            )


def stream_of(t):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_MULTI = None


def device_ctx(t):
    """`with torch.cuda.device(...)` only when several GPUs are visible (utils.py:170-178)."""
    global _MULTI
    if _MULTI is None:
        _MULTI = torch.cuda.device_count() > 1
    return torch.cuda.device(t.device) if _MULTI else nullcontext()


def check(rc, name):
    if rc != 0:
        why = _ERR.get(rc, f"hipError {rc}")
        raise RuntimeError(f"unsloth_amd: {name} failed: {why} (code {rc})")
