"""QuickGELU (x * sigmoid(1.702 x)) through the HIP kernels (csrc/glu.hip: uamd_quick_gelu_forward / _backward): the activation
of Qwen2-VL's vision MLP (BASELINE config 4). The reference's VLM path compiles HF's module tree (unsloth/models/vision.py:
881-1990, unsloth_zoo compiler -- third party); the semantics restated here are transformers' QuickGELUActivation. The backward
writes dX IN PLACE over dY, like the SwiGLU / GeGLU backward kernels of the language tower."""
import torch

from .. import _lib


class Fast_QuickGELU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        _lib.require_gpu(X)
        Xc = X if X.is_contiguous() else X.contiguous()
        Y = torch.empty_like(Xc)
        with _lib.device_ctx(Xc):
            rc = _lib.lib().uamd_quick_gelu_forward(_lib.ptr(Xc), _lib.ptr(Y), Xc.numel(), _lib.dtype_code(Xc.dtype), _lib.stream_of(Xc))
        _lib.check(rc, "uamd_quick_gelu_forward")
        ctx.save_for_backward(Xc)
        return Y.view(X.shape)

    @staticmethod
    def backward(ctx, dY):
        (X,) = ctx.saved_tensors
        d = dY if (dY.is_contiguous() and dY.dtype == X.dtype) else dY.to(X.dtype).contiguous()
        with _lib.device_ctx(d):
            rc = _lib.lib().uamd_quick_gelu_backward(_lib.ptr(X), _lib.ptr(d), X.numel(), _lib.dtype_code(X.dtype), _lib.stream_of(d))
        _lib.check(rc, "uamd_quick_gelu_backward")
        return d.view(dY.shape)


def fast_quick_gelu(X):
    return Fast_QuickGELU.apply(X)
