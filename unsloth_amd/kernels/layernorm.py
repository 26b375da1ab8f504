"""LayerNorm through the HIP kernels (csrc/layernorm.hip); mirror of unsloth/kernels/layernorm.py.

  Fast_Layernorm   (:107-163)  forward saves (X, W, b, r, mu); backward writes dX over dY and returns no dW / db
  fast_layernorm   (:166-173)  module-level entry: `fast_layernorm(layernorm, X)`
  patch_layernorm              the reference imports it from unsloth_zoo.patching_utils (:20-22; third party, not in
                               the repository): here it routes torch.nn.LayerNorm.forward through fast_layernorm for
                               GPU tensors with an affine weight + bias (the vision towers' norms), else torch's own.
All arithmetic in fp32 with one rounding to X's dtype, like the reference ("all modules are in float32", :47-49)."""
import torch

from .. import _lib


class Fast_Layernorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, W, b, eps):
        _lib.require_gpu(X, W, b)
        shape = X.shape
        dim = shape[-1]
        X2 = X.reshape(-1, dim)
        if X2.stride(1) != 1:
            X2 = X2.contiguous()
        n_rows = X2.shape[0]
        Wc = W if W.is_contiguous() else W.contiguous()
        bc = b if (b.is_contiguous() and b.dtype == Wc.dtype) else b.to(Wc.dtype).contiguous()
        Y = torch.empty((n_rows, dim), dtype=X.dtype, device=X.device)
        r = torch.empty(n_rows, dtype=torch.float32, device=X.device)
        mu = torch.empty(n_rows, dtype=torch.float32, device=X.device)
        with _lib.device_ctx(X2):
            rc = _lib.lib().uamd_layernorm_fwd(_lib.ptr(X2), _lib.ptr(Wc), _lib.ptr(bc), _lib.ptr(Y), _lib.ptr(r),
                                               _lib.ptr(mu), n_rows, dim, X2.stride(0), Y.stride(0), float(eps),
                                               _lib.dtype_code(X.dtype), _lib.dtype_code(Wc.dtype), _lib.stream_of(X2))
        _lib.check(rc, "uamd_layernorm_fwd")
        ctx.save_for_backward(X2, Wc, r, mu)
        return Y.view(*shape)

    @staticmethod
    def backward(ctx, dY):
        X2, W, r, mu = ctx.saved_tensors
        shape = dY.shape
        dim = shape[-1]
        dY2 = dY.reshape(-1, dim)
        if dY2.stride(1) != 1 or dY2.dtype != X2.dtype:
            dY2 = dY2.to(X2.dtype).contiguous()
        with _lib.device_ctx(dY2):
            rc = _lib.lib().uamd_layernorm_bwd(_lib.ptr(dY2), _lib.ptr(X2), _lib.ptr(W), _lib.ptr(r), _lib.ptr(mu),
                                               dY2.shape[0], dim, dY2.stride(0), X2.stride(0),
                                               _lib.dtype_code(dY2.dtype), _lib.dtype_code(W.dtype), _lib.stream_of(dY2))
        _lib.check(rc, "uamd_layernorm_bwd")
        return dY2.view(*shape), None, None, None          # dX written over dY (layernorm.py:104, :161-163)


@torch.compiler.disable
def fast_layernorm(layernorm, X):
    """layernorm.py:166-173."""
    assert layernorm.elementwise_affine is True
    eps = layernorm.variance_epsilon if hasattr(layernorm, "variance_epsilon") else layernorm.eps
    return Fast_Layernorm.apply(X, layernorm.weight, layernorm.bias, eps)


_ORIGINAL_FORWARD = [None]


def _supported(module, X):
    W, b = getattr(module, "weight", None), getattr(module, "bias", None)
    return (X.is_cuda and W is not None and b is not None and len(module.normalized_shape) == 1
            and X.dtype in (torch.bfloat16, torch.float16, torch.float32) and W.dtype in (X.dtype, torch.float32)
            and not W.requires_grad and not b.requires_grad)


def patch_layernorm():
    """torch.nn.LayerNorm.forward -> fast_layernorm where it applies (frozen affine norm over the last dimension on the
    GPU: the reference's kernel returns no dW / db), torch's own forward otherwise. Idempotent."""
    if _ORIGINAL_FORWARD[0] is not None:
        return
    original = torch.nn.LayerNorm.forward
    _ORIGINAL_FORWARD[0] = original

    def forward(self, X):
        if _supported(self, X):
            return fast_layernorm(self, X)
        return original(self, X)
    torch.nn.LayerNorm.forward = forward


def unpatch_layernorm():
    if _ORIGINAL_FORWARD[0] is not None:
        torch.nn.LayerNorm.forward = _ORIGINAL_FORWARD[0]
        _ORIGINAL_FORWARD[0] = None
