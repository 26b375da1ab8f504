"""Manual-autograd blocks for TRAINABLE dense projections: full fine-tuning (BASELINE config 3).

The reference routes `full_finetuning=True` away from its hand kernels (loader.py:487-523 -> FastModel -> the
unsloth_zoo compiler): the projections are plain torch.nn.Linear modules under autograd. Here the same three blocks as
kernels/fast_lora.py -- fused Q|K|V, the gated MLP, a single projection -- with the LoRA terms gone and the weight
gradient in their place:
    Y  = X W^T                              uamd_gemm_nt_256 (grouped over the projections that share X)
    dX = sum_g dY_g W_g                     uamd_gemm_nn_256 (contracts over W's rows as stored)
    dW_g = dY_g^T X                         uamd_gemm_tn_256 (both operands read where the backward left them)
dW is written straight into the parameter's gradient sink when it has one (full_finetune.FullGradBuckets: flat per-layer
buckets that are reduce-scattered while the next layer's backward runs), else returned to autograd. The gated
activation and its in-place backward are the kernels of kernels/swiglu.py / geglu.py (the reference's swiglu.py:33-143).
"""
import torch

from .fast_lora import _bias_grad, _custom_bwd, _custom_fwd
from .utils import _rows2d, dense_dw, grad_sink, lora_linear_dx, lora_linear_forward


def _proj(W, bias=None):
    return (W, None, None, None, None, bias)


def _adjacent_rows(views):
    """2-D contiguous views that are consecutive ROW blocks of one buffer -> the stacked [sum rows, cols] view, else None."""
    v0 = views[0]
    off = 0
    for v in views:
        if (v.dim() != 2 or not v.is_contiguous() or v.shape[1] != v0.shape[1] or v.dtype != v0.dtype
                or v.data_ptr() != v0.data_ptr() + off * v0.element_size()):
            return None
        off += v.numel()
    return torch.as_strided(v0, (off // v0.shape[1], v0.shape[1]), (v0.shape[1], 1))


def weight_grads(dYs, X, Ws, needed):
    """[dW_g] for projections sharing the input X (None where not needed / where the gradient went into a sink).
    When every projection needs its gradient, the dY_g are column blocks of one buffer (the attention backward writes
    dQ | dK | dV that way) and the destinations are row blocks of one buffer (the flat gradient bucket), the whole set is
    ONE GEMM: k_proj / v_proj alone (1024 x 4096 = 64 tiles) would leave three quarters of the chip idle."""
    from .utils import _adjacent_columns
    X2 = _rows2d(X)
    dY2 = [_rows2d(d) for d in dYs]
    sinks = [grad_sink(W) if n else None for W, n in zip(Ws, needed)]
    out = [None] * len(Ws)
    if len(Ws) > 1 and all(needed):
        dcat = _adjacent_columns(dY2)
        if dcat is not None and dcat.stride(0) % 8 == 0 and dcat.data_ptr() % 16 == 0:
            if all(s is not None for s in sinks):
                views = [s.grad_view(W) for s, W in zip(sinks, Ws)]
                firsts = [s.first_write(W) for s, W in zip(sinks, Ws)]
                stacked = _adjacent_rows(views)
                if stacked is not None and len(set(firsts)) == 1:
                    dense_dw(dcat, X2, out=stacked, accumulate=not firsts[0])
                    for s, W in zip(sinks, Ws):
                        s.ready(W)
                    return out
            elif all(s is None for s in sinks):
                full = dense_dw(dcat, X2)
                r0 = 0
                for i, W in enumerate(Ws):
                    out[i] = full[r0:r0 + W.shape[0]]
                    r0 += W.shape[0]
                return out
    for i, (d, W, n, s) in enumerate(zip(dY2, Ws, needed, sinks)):
        if not n:
            continue
        if s is not None:
            first = s.first_write(W)
            dense_dw(d, X2, out=s.grad_view(W), accumulate=not first)
            s.ready(W)
        else:
            out[i] = dense_dw(d, X2)
    return out


def _bias_grads(dYs, biases, needed):
    return tuple(_bias_grad(d, b, n) for d, b, n in zip(dYs, biases, needed))


class Dense_QKV(torch.autograd.Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, X, QW, KW, VW, Qb, Kb, Vb, inplace=True):
        Q, K, V = lora_linear_forward(X, [_proj(QW, Qb), _proj(KW, Kb), _proj(VW, Vb)])
        ctx.save_for_backward(X, QW, KW, VW)
        ctx.biases = (Qb, Kb, Vb)
        ctx.inplace = inplace
        return Q, K, V

    @staticmethod
    @_custom_bwd
    def backward(ctx, dQ, dK, dV):
        X, QW, KW, VW = ctx.saved_tensors
        nig = ctx.needs_input_grad
        dYs = [d.reshape(-1, d.shape[-1]) for d in (dQ, dK, dV)]
        dWs = weight_grads(dYs, X, (QW, KW, VW), nig[1:4])           # BEFORE dX overwrites X
        dbs = _bias_grads(dYs, ctx.biases, nig[4:7])
        dX = None
        if nig[0]:
            X2 = X.reshape(-1, X.shape[-1])
            dX = lora_linear_dx(dYs, [_proj(QW), _proj(KW), _proj(VW)],
                                out=X2 if (ctx.inplace and X2.is_contiguous()) else None).view(X.shape)
        return (dX,) + tuple(dWs) + dbs + (None,)


class Dense_W(torch.autograd.Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, X, W, bias=None):
        (Y,) = lora_linear_forward(X, [_proj(W, bias)])
        ctx.save_for_backward(X, W)
        ctx.bias = bias
        return Y

    @staticmethod
    @_custom_bwd
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        nig = ctx.needs_input_grad
        dY2 = dY.reshape(-1, dY.shape[-1])
        (dW,) = weight_grads([dY2], X, (W,), nig[1:2])
        db = _bias_grad(dY2, ctx.bias, len(nig) > 2 and nig[2])
        dX = lora_linear_dx([dY2], [_proj(W)]).view(X.shape) if nig[0] else None
        return dX, dW, db


class Dense_MLP(torch.autograd.Function):
    """out = act(X Wg^T) * (X Wu^T) Wd^T. Saves X, e, g (the reference's LoRA_MLP saves the same three, fast_lora.py:
    93-125); the backward rebuilds h in place of dh like swiglu_DWf_DW_dfg_kernel does (swiglu.py:79-143)."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, X, gateW, upW, downW, act_fwd, act_bwd, inplace=True):
        e, g = lora_linear_forward(X, [_proj(gateW), _proj(upW)])
        h = act_fwd(e, g)
        (out,) = lora_linear_forward(h, [_proj(downW)])
        ctx.save_for_backward(X, e, g, gateW, upW, downW)
        ctx.act_bwd = act_bwd
        ctx.inplace = inplace
        return out

    @staticmethod
    @_custom_bwd
    def backward(ctx, dY):
        X, e, g, gateW, upW, downW = ctx.saved_tensors
        nig = ctx.needs_input_grad
        dY2 = dY.reshape(-1, dY.shape[-1])
        X2 = X.reshape(-1, X.shape[-1])
        e2, g2 = e.view(-1, e.shape[-1]), g.view(-1, g.shape[-1])
        DW = lora_linear_dx([dY2], [_proj(downW)])                   # dh = dY @ W_down          (fast_lora.py:156)
        h, df, de = ctx.act_bwd(DW, e2, g2)                          # in place: DW -> h, e -> df, g -> de   (:157)
        (d_down,) = weight_grads([dY2], h, (downW,), nig[3:4])
        d_up, d_gate = weight_grads([df, de], X2, (upW, gateW), (nig[2], nig[1]))
        dX = None
        if nig[0]:
            dX = lora_linear_dx([df, de], [_proj(upW), _proj(gateW)],
                                out=X2 if (ctx.inplace and X2.is_contiguous()) else None).view(X.shape)
        return dX, d_gate, d_up, d_down, None, None, None


# ---- hooks installed on the HF modules by models/full_finetune.py ---------------------------------------------------
def apply_dense_qkv(self, X, inplace=True):
    q, k, v = self.q_proj, self.k_proj, self.v_proj
    return Dense_QKV.apply(X, q.weight, k.weight, v.weight, q.bias, k.bias, v.bias, inplace)


def apply_dense_o(self, X):
    return Dense_W.apply(X, self.o_proj.weight, self.o_proj.bias)


def apply_dense_mlp_swiglu(self, X, inplace=True):
    from .swiglu import swiglu_DWf_DW_dfg_kernel, swiglu_fg_kernel
    return Dense_MLP.apply(X, self.gate_proj.weight, self.up_proj.weight, self.down_proj.weight, swiglu_fg_kernel,
                           swiglu_DWf_DW_dfg_kernel, inplace)
