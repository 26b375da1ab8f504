"""SwiGLU forward / fused in-place backward; mirror of unsloth/kernels/swiglu.py:50-64,112-125."""
import torch

from .. import _lib


def _glu_fwd(name, e, g):
    _lib.require_gpu(e, g)
    assert e.shape == g.shape and e.dtype == g.dtype
    e, g = e.contiguous(), g.contiguous()
    h = torch.empty_like(e)
    with _lib.device_ctx(e):
        rc = getattr(_lib.lib(), name)(_lib.ptr(e), _lib.ptr(g), _lib.ptr(h), e.numel(),
                                       _lib.dtype_code(e.dtype), _lib.stream_of(e))
    _lib.check(rc, name)
    return h


def _glu_bwd(name, DW, e, g):
    """In place: DW <- h, e <- df, g <- de. The three buffers must be contiguous (they are the
    GEMM outputs saved by LoRA_MLP); a non-contiguous input cannot honour the aliasing contract."""
    _lib.require_gpu(DW, e, g)
    if not (DW.is_contiguous() and e.is_contiguous() and g.is_contiguous()):
        raise ValueError("in-place GLU backward needs contiguous DW, e, g")
    assert DW.shape == e.shape == g.shape and DW.dtype == e.dtype == g.dtype
    with _lib.device_ctx(e):
        rc = getattr(_lib.lib(), name)(_lib.ptr(DW), _lib.ptr(e), _lib.ptr(g), e.numel(),
                                       _lib.dtype_code(e.dtype), _lib.stream_of(e))
    _lib.check(rc, name)
    return DW, e, g


def swiglu_fg_kernel(e, g):
    """h = (e * sigmoid(e)).to(dtype) * g   (swiglu.py:27-64)."""
    return _glu_fwd("uamd_swiglu_fg", e, g)


def swiglu_DWf_DW_dfg_kernel(DW, e, g):
    """(h, df, de) written over (DW, e, g)   (swiglu.py:67-125)."""
    return _glu_bwd("uamd_swiglu_DWf_DW_dfg", DW, e, g)
