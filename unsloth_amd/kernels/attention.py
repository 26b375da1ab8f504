"""Causal GQA flash attention through the HIP kernels (csrc/attention.hip).

The reference delegates this step to flash-attn / xformers / torch SDPA (`run_attention`,
unsloth/utils/attention_dispatch.py:298-617, called from unsloth/models/llama.py:757); SURVEY 8(f1). Here it is a
pair of hand-written CDNA4 kernels that read Q/K/V in the [B, T, H, D] layout the QKV GEMM produced (any strides,
d contiguous) and write O as [B, T, Hq, D] == the o_proj input, so none of the reference's transposes
(llama.py:276-277) or `.contiguous()` copies exist.
"""
import ctypes
import math

import torch

from .. import _lib

_I64x12 = ctypes.c_int64 * 12
_I64x24 = ctypes.c_int64 * 24


def supported(q, k, v):
    return (q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and q.shape[-1] == 128
            and k.shape[2] > 0 and q.shape[2] % k.shape[2] == 0 and (q.shape[2] // k.shape[2]) in (1, 2, 4, 8))


def _strides(*ts):
    out = []
    for t in ts:
        assert t.stride(3) == 1, "head_dim must be contiguous"
        out += [t.stride(0), t.stride(1), t.stride(2)]
    return (_I64x12 if len(ts) == 4 else _I64x24)(*out)


def _pad32(T):
    return (T + 31) // 32 * 32


def attn_forward(q, k, v, scale=None):
    """q [B,T,Hq,128], k/v [B,T,Hk,128] (strided views are fine) -> (o [B,T,Hq,128] contiguous, lse [B,Hq,T] fp32)."""
    _lib.require_gpu(q, k, v)
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    o = torch.empty((B, T, Hq, D), dtype=q.dtype, device=q.device)
    Tp = _pad32(T)
    lse = (torch.empty if Tp == T else torch.zeros)((B, Hq, Tp), dtype=torch.float32, device=q.device)
    with _lib.device_ctx(q):
        rc = _lib.lib().uamd_attn_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(lse),
                                      _strides(q, k, v, o), B, T, Hq, Hk, D, Tp, float(scale), 1,
                                      _lib.dtype_code(q.dtype), _lib.stream_of(q))
    _lib.check(rc, "uamd_attn_fwd")
    return o, lse[:, :, :T]


def attn_backward(do, q, k, v, o, lse, scale=None):
    """Gradients of attn_forward: (dq [B,T,Hq,D], dk, dv [B,T,Hk,D]), contiguous, in q's dtype. `lse` is the view
    attn_forward returned (its storage is padded to a multiple of 32 positions). Two launches, deterministic."""
    _lib.require_gpu(do, q, k, v, o)
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    Tp = _pad32(T)
    assert lse.stride(1) == Tp and lse.stride(2) == 1, "pass the LSE returned by attn_forward"
    if do.stride(3) != 1:
        do = do.contiguous()
    dq = torch.empty((B, T, Hq, D), dtype=q.dtype, device=q.device)
    dk = torch.empty((B, T, Hk, D), dtype=q.dtype, device=q.device)
    dv = torch.empty((B, T, Hk, D), dtype=q.dtype, device=q.device)
    delta = (torch.empty if Tp == T else torch.zeros)((B, Hq, Tp), dtype=torch.float32, device=q.device)
    with _lib.device_ctx(q):
        rc = _lib.lib().uamd_attn_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(do),
                                      _lib.ptr(lse), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(delta),
                                      _strides(q, k, v, o, do, dq, dk, dv), B, T, Hq, Hk, D, Tp, float(scale), 1,
                                      _lib.dtype_code(q.dtype), _lib.stream_of(q))
    _lib.check(rc, "uamd_attn_bwd")
    return dq, dk, dv


class FlashAttention(torch.autograd.Function):
    """o = causal_attention(q, k, v) on [B,T,H,128] views; saves (q, k, v, o, lse) like flash-attention."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        o, lse = attn_forward(q, k, v, scale)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = attn_backward(do, q, k, v, o, lse, ctx.scale)
        return dq, dk, dv, None


def flash_attention(q, k, v, scale=None):
    return FlashAttention.apply(q, k, v, scale)
