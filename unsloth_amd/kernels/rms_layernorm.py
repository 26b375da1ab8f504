"""RMSNorm through the HIP kernels; mirror of unsloth/kernels/rms_layernorm.py.

Same autograd contract as the reference's Fast_RMS_Layernorm (:162-240): saves (X, W, r); the
backward writes dX IN PLACE over dY for the non-Gemma case (:92-95, :218) and returns no dW
(norm weights are frozen under LoRA).
"""
import torch

from .. import _lib


class Fast_RMS_Layernorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, W, eps, gemma=False):
        _lib.require_gpu(X, W)
        shape = X.shape
        dim = shape[-1]
        X = X.reshape(-1, dim)
        if X.stride(1) != 1:
            X = X.contiguous()
        n_rows, n_cols = X.shape
        Y = torch.empty((n_rows, n_cols), dtype=X.dtype, device=X.device)
        r = torch.empty(n_rows, dtype=torch.float32, device=X.device)
        W = W.contiguous()
        with _lib.device_ctx(X):
            rc = _lib.lib().uamd_rms_layernorm_fwd(
                _lib.ptr(X), _lib.ptr(W), _lib.ptr(Y), _lib.ptr(r), n_rows, n_cols, X.stride(0),
                Y.stride(0), float(eps), int(bool(gemma)), _lib.dtype_code(X.dtype),
                _lib.dtype_code(W.dtype), _lib.stream_of(X))
        _lib.check(rc, "uamd_rms_layernorm_fwd")
        ctx.eps = eps
        ctx.GEMMA = bool(gemma)
        ctx.save_for_backward(X, W, r)
        return Y.view(*shape)

    @staticmethod
    def backward(ctx, dY):
        shape = dY.shape
        dim = shape[-1]
        dY = dY.reshape(-1, dim)
        if dY.stride(1) != 1:
            dY = dY.contiguous()
        X, W, r = ctx.saved_tensors
        n_rows, n_cols = dY.shape
        dX = torch.empty_like(dY) if ctx.GEMMA else dY      # rms_layernorm.py:218
        with _lib.device_ctx(dY):
            rc = _lib.lib().uamd_rms_layernorm_bwd(
                _lib.ptr(dY), _lib.ptr(dX), _lib.ptr(X), _lib.ptr(W), _lib.ptr(r), n_rows, n_cols,
                dY.stride(0), dX.stride(0), X.stride(0), int(ctx.GEMMA), _lib.dtype_code(dY.dtype),
                _lib.dtype_code(W.dtype), _lib.stream_of(dY))
        _lib.check(rc, "uamd_rms_layernorm_bwd")
        return dX.view(*shape), None, None, None


@torch.compiler.disable
def fast_rms_layernorm(layernorm, X, gemma=False):
    """rms_layernorm.py:244-255."""
    W = layernorm.weight
    eps = layernorm.variance_epsilon if hasattr(layernorm, "variance_epsilon") else layernorm.eps
    return Fast_RMS_Layernorm.apply(X, W, eps, gemma)


from transformers.models.llama.modeling_llama import LlamaRMSNorm


class Unsloth_LlamaRMSNorm(LlamaRMSNorm):
    def forward(self, X):
        return fast_rms_layernorm(self, X, gemma=False)


def patch_rms_layernorm():
    """rms_layernorm.py:277-286: swap the HF class so newly built models use the fast norm."""
    import transformers.models.llama.modeling_llama as m
    m.LlamaRMSNorm = Unsloth_LlamaRMSNorm


def unpatch_rms_layernorm():
    import transformers.models.llama.modeling_llama as m
    m.LlamaRMSNorm = LlamaRMSNorm
