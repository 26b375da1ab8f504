"""RMSNorm through the HIP kernels; mirror of unsloth/kernels/rms_layernorm.py.

Same autograd contract as the reference's Fast_RMS_Layernorm (:162-240): saves (X, W, r); the
backward writes dX IN PLACE over dY for the non-Gemma case (:92-95, :218). The reference returns no dW
(norm weights are frozen under LoRA; with full_finetuning=True it leaves trainable norms to HF's torch module).
Here a weight that requires grad gets it from `uamd_rms_layernorm_dw` (csrc/rms_layernorm.hip), computed from dY
before the in-place dX pass overwrites it.
"""
import torch

from .. import _lib


# ---- plain launchers (no autograd), shared by the Functions below and by models/fast_layer.py ---------------------
def _rows(X):
    X2 = X.reshape(-1, X.shape[-1])
    return X2 if X2.stride(1) == 1 else X2.contiguous()


def rms_fwd(X, W, eps, gemma=False):
    """(Y, r) for X [..., dim]; Y [rows, dim]."""
    _lib.require_gpu(X, W)
    X2 = _rows(X)
    n_rows, n_cols = X2.shape
    Y = torch.empty((n_rows, n_cols), dtype=X2.dtype, device=X2.device)
    r = torch.empty(n_rows, dtype=torch.float32, device=X2.device)
    W = W.contiguous()
    with _lib.device_ctx(X2):
        rc = _lib.lib().uamd_rms_layernorm_fwd(
            _lib.ptr(X2), _lib.ptr(W), _lib.ptr(Y), _lib.ptr(r), n_rows, n_cols, X2.stride(0), Y.stride(0),
            float(eps), int(bool(gemma)), _lib.dtype_code(X2.dtype), _lib.dtype_code(W.dtype), _lib.stream_of(X2))
    _lib.check(rc, "uamd_rms_layernorm_fwd")
    return Y, r


def add_rms_fwd(X, residual, W, eps):
    """(H, Y, r): H = X + residual, Y = rmsnorm(H) * W."""
    _lib.require_gpu(X, residual, W)
    X2, R2 = _rows(X), _rows(residual)
    n_rows, dim = X2.shape
    H = torch.empty((n_rows, dim), dtype=X2.dtype, device=X2.device)
    Y = torch.empty((n_rows, dim), dtype=X2.dtype, device=X2.device)
    r = torch.empty(n_rows, dtype=torch.float32, device=X2.device)
    W = W.contiguous()
    with _lib.device_ctx(X2):
        rc = _lib.lib().uamd_add_rms_layernorm_fwd(
            _lib.ptr(X2), _lib.ptr(R2), _lib.ptr(W), _lib.ptr(H), _lib.ptr(Y), _lib.ptr(r), n_rows, dim,
            X2.stride(0), R2.stride(0), H.stride(0), Y.stride(0), float(eps), _lib.dtype_code(X2.dtype),
            _lib.dtype_code(W.dtype), _lib.stream_of(X2))
    _lib.check(rc, "uamd_add_rms_layernorm_fwd")
    return H, Y, r


def rms_dw(dY, X, r, W, out=None, accumulate=False):
    """dW[c] (+)= sum_rows dY[row, c] * X[row, c] * r[row] in W's dtype (X = the norm's input). Deterministic."""
    from .. import nf4 as _nf4
    dY2, X2 = _rows(dY), _rows(X)
    n_rows, dim = dY2.shape
    if out is None:
        out = torch.empty(dim, dtype=W.dtype, device=W.device)
        accumulate = False
    assert out.is_contiguous() and out.numel() == dim and out.dtype == W.dtype
    col_blocks = (dim // (16 // dY2.element_size()) + 255) // 256
    chunks = max(1, min((2048 + col_blocks - 1) // col_blocks, (n_rows + 7) // 8))
    ws = _nf4.scratch(dY2.device, chunks * dim, torch.float32, slot=41)
    with _lib.device_ctx(dY2):
        rc = _lib.lib().uamd_rms_layernorm_dw(
            _lib.ptr(dY2), _lib.ptr(X2), _lib.ptr(r), _lib.ptr(out), _lib.ptr(ws), ws.numel(), n_rows, dim,
            dY2.stride(0), X2.stride(0), int(bool(accumulate)), _lib.dtype_code(dY2.dtype), _lib.dtype_code(W.dtype),
            _lib.stream_of(dY2))
    _lib.check(rc, "uamd_rms_layernorm_dw")
    return out


def _weight_grad(dY, X, r, W, needed):
    """The norm weight's gradient for autograd (None when the weight is frozen): added straight into the parameter's
    gradient sink when it has one (full fine-tuning's flat gradient buckets), else returned."""
    if not needed:
        return None
    from .utils import grad_sink
    sink = grad_sink(W)
    if sink is not None:
        first = sink.first_write(W) if hasattr(sink, "first_write") else False
        rms_dw(dY, X, r, W, out=sink.grad_view(W).view(-1), accumulate=not first)
        sink.ready(W)
        return None
    return rms_dw(dY, X, r, W)


def rms_bwd_(dY, H, W, r, dH=None):
    """dX = rmsnorm_backward(dY; H, W, r) (+ dH, the gradient reaching H from the residual path), written IN PLACE
    over dY (rms_layernorm.py:218) and returned. Llama-style norm."""
    dY2 = _rows(dY)
    n_rows, dim = dY2.shape
    H2 = _rows(H)
    with _lib.device_ctx(dY2):
        if dH is None:
            rc = _lib.lib().uamd_rms_layernorm_bwd(
                _lib.ptr(dY2), _lib.ptr(dY2), _lib.ptr(H2), _lib.ptr(W), _lib.ptr(r), n_rows, dim, dY2.stride(0),
                dY2.stride(0), H2.stride(0), 0, _lib.dtype_code(dY2.dtype), _lib.dtype_code(W.dtype),
                _lib.stream_of(dY2))
            _lib.check(rc, "uamd_rms_layernorm_bwd")
        else:
            dH2 = _rows(dH)
            rc = _lib.lib().uamd_add_rms_layernorm_bwd(
                _lib.ptr(dY2), _lib.ptr(dH2), _lib.ptr(dY2), _lib.ptr(H2), _lib.ptr(W), _lib.ptr(r), n_rows, dim,
                dY2.stride(0), dH2.stride(0), dY2.stride(0), H2.stride(0), _lib.dtype_code(dY2.dtype),
                _lib.dtype_code(W.dtype), _lib.stream_of(dY2))
            _lib.check(rc, "uamd_add_rms_layernorm_bwd")
    return dY2


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
        dW = _weight_grad(dY, X, r, W, ctx.needs_input_grad[1])
        dX = torch.empty_like(dY) if ctx.GEMMA else dY      # rms_layernorm.py:218
        with _lib.device_ctx(dY):
            rc = _lib.lib().uamd_rms_layernorm_bwd(
                _lib.ptr(dY), _lib.ptr(dX), _lib.ptr(X), _lib.ptr(W), _lib.ptr(r), n_rows, n_cols,
                dY.stride(0), dX.stride(0), X.stride(0), int(ctx.GEMMA), _lib.dtype_code(dY.dtype),
                _lib.dtype_code(W.dtype), _lib.stream_of(dY))
        _lib.check(rc, "uamd_rms_layernorm_bwd")
        return dX.view(*shape), dW, None, None


class Fast_Add_RMS_Layernorm(torch.autograd.Function):
    """(h, y) = (X + residual, rmsnorm(X + residual) * W) in one pass over the activations; the backward adds the
    gradient that reaches h from the residual path inside the norm's backward kernel. Same numbers as
    `h = residual + X; y = Fast_RMS_Layernorm(h)` (llama.py:823-844) with the autograd accumulation of dh."""

    @staticmethod
    def forward(ctx, X, residual, W, eps):
        _lib.require_gpu(X, residual, W)
        shape = X.shape
        dim = shape[-1]
        X2 = X.reshape(-1, dim)
        R2 = residual.reshape(-1, dim)
        if X2.stride(1) != 1:
            X2 = X2.contiguous()
        if R2.stride(1) != 1:
            R2 = R2.contiguous()
        n_rows = X2.shape[0]
        H = torch.empty((n_rows, dim), dtype=X.dtype, device=X.device)
        Y = torch.empty((n_rows, dim), dtype=X.dtype, device=X.device)
        r = torch.empty(n_rows, dtype=torch.float32, device=X.device)
        W = W.contiguous()
        with _lib.device_ctx(X):
            rc = _lib.lib().uamd_add_rms_layernorm_fwd(
                _lib.ptr(X2), _lib.ptr(R2), _lib.ptr(W), _lib.ptr(H), _lib.ptr(Y), _lib.ptr(r), n_rows, dim,
                X2.stride(0), R2.stride(0), H.stride(0), Y.stride(0), float(eps), _lib.dtype_code(X.dtype),
                _lib.dtype_code(W.dtype), _lib.stream_of(X))
        _lib.check(rc, "uamd_add_rms_layernorm_fwd")
        ctx.set_materialize_grads(False)           # an unused h (last layer) arrives as dH = None, not as a zero tensor
        ctx.save_for_backward(H, W, r)
        return H.view(*shape), Y.view(*shape)

    @staticmethod
    def backward(ctx, dH, dY):
        H, W, r = ctx.saved_tensors
        if dY is None:                             # only the residual stream was used downstream
            return dH, dH, None, None
        shape = dY.shape
        dim = shape[-1]
        dY = dY.reshape(-1, dim)
        if dY.stride(1) != 1:
            dY = dY.contiguous()
        n_rows = dY.shape[0]
        dW = _weight_grad(dY, H, r, W, ctx.needs_input_grad[2])
        with _lib.device_ctx(dY):
            if dH is None:
                rc = _lib.lib().uamd_rms_layernorm_bwd(
                    _lib.ptr(dY), _lib.ptr(dY), _lib.ptr(H), _lib.ptr(W), _lib.ptr(r), n_rows, dim, dY.stride(0),
                    dY.stride(0), H.stride(0), 0, _lib.dtype_code(dY.dtype), _lib.dtype_code(W.dtype),
                    _lib.stream_of(dY))
            else:
                dH = dH.reshape(-1, dim)
                if dH.stride(1) != 1:
                    dH = dH.contiguous()
                rc = _lib.lib().uamd_add_rms_layernorm_bwd(
                    _lib.ptr(dY), _lib.ptr(dH), _lib.ptr(dY), _lib.ptr(H), _lib.ptr(W), _lib.ptr(r), n_rows, dim,
                    dY.stride(0), dH.stride(0), dY.stride(0), H.stride(0), _lib.dtype_code(dY.dtype),
                    _lib.dtype_code(W.dtype), _lib.stream_of(dY))
        _lib.check(rc, "uamd_add_rms_layernorm_bwd")
        dX = dY.view(*shape)                       # written over dY, like rms_layernorm.py:218
        return dX, dX, dW, None


def add_rms_supported(X, W):
    """shapes the fused kernel takes (otherwise: torch add + fast_rms_layernorm)."""
    vec = 16 // X.element_size()
    return (X.is_cuda and X.dtype in (torch.bfloat16, torch.float16, torch.float32) and X.shape[-1] % vec == 0
            and X.shape[-1] <= 64 * vec * 8 and W.dtype in (X.dtype, torch.float32))


@torch.compiler.disable
def fast_add_rms_layernorm(layernorm, X, residual):
    """(residual + X, layernorm(residual + X)) -- the add of llama.py:833/:840 fused into the following norm."""
    W = layernorm.weight
    eps = layernorm.variance_epsilon if hasattr(layernorm, "variance_epsilon") else layernorm.eps
    if not add_rms_supported(X, W):
        h = residual + X
        return h, Fast_RMS_Layernorm.apply(h, W, eps, False)
    return Fast_Add_RMS_Layernorm.apply(X, residual, W, eps)


@torch.compiler.disable
def fast_rms_layernorm(layernorm, X, gemma=False):
    """rms_layernorm.py:244-255."""
    W = layernorm.weight
    eps = layernorm.variance_epsilon if hasattr(layernorm, "variance_epsilon") else layernorm.eps
    return Fast_RMS_Layernorm.apply(X, W, eps, gemma)


from transformers.models.llama.modeling_llama import LlamaRMSNorm


class Unsloth_LlamaRMSNorm(LlamaRMSNorm):
    def forward(self, X):
        if not X.is_cuda:                        # a stock model on the host (e.g. an fp32 reference): HF's own forward
            return super().forward(X)
        return fast_rms_layernorm(self, X, gemma=False)


def patch_rms_layernorm():
    """rms_layernorm.py:277-286: swap the HF class so newly built models use the fast norm."""
    import transformers.models.llama.modeling_llama as m
    m.LlamaRMSNorm = Unsloth_LlamaRMSNorm


def unpatch_rms_layernorm():
    import transformers.models.llama.modeling_llama as m
    m.LlamaRMSNorm = LlamaRMSNorm
