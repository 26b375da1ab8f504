"""Manual-autograd LoRA blocks; mirror of unsloth/kernels/fast_lora.py.

  LoRA_MLP  (:28-229)   e = X Wg^T(+LoRA), g = X Wu^T(+LoRA), h = act(e,g), i = h Wd^T(+LoRA)
  LoRA_QKV  (:335-540)  Q,K,V = X W{q,k,v}^T (+LoRA)
  LoRA_W    (:574-650)  single projection (o_proj)
  apply_lora_mlp_swiglu / _geglu_exact / _geglu_approx (:235-332), apply_lora_qkv (:543-571),
  apply_lora_o (:653-657), fast_lora_forward (:662-665, raises NotImplementedError like the reference)

Same saved tensors, same in-place contracts (the activation backward overwrites DW/e/g; dX is
written into the saved X buffer when inplace=True, :193-204, :497-517), same gradient formulas:
    dA = s * (dY B)^T X      dB = s * dY^T (X A^T)      dX = dY W + s (dY B) A
What changes is the launch structure (kernels/utils.py): gate+up and q+k+v are one grouped MFMA GEMM
launch each, the LoRA term rides in that launch as extra K tiles (rank block), every dX is one
transposing-dequant launch + one GEMM (q/k/v: ONE K-concatenated GEMM), and all rank-r gradient
products of a block are ONE `uamd_lora_tn` launch (csrc/lora_side.hip) that can add straight into
the data-parallel gradient arena. Each block also exists as a pair of plain functions
(`mlp_forward` / `mlp_backward`, ...) for the whole-layer Function with selective recompute
(models/fast_layer.py).
"""
import os

import torch

from .utils import _same_rows, alloc_rows, glu_bwd_terms, glu_fwd_xa
from .utils import (
    GRAD_SINKS,
    grad_sink,
    cast_lora,
    get_lora_parameters,
    lora_dx_terms,
    lora_linear_dx,
    lora_linear_forward,
    lora_tn,
    lora_tn_supported,
)
from .swiglu import swiglu_fg_kernel, swiglu_DWf_DW_dfg_kernel
from .geglu import (
    geglu_exact_forward_kernel,
    geglu_exact_backward_kernel,
    geglu_approx_forward_kernel,
    geglu_approx_backward_kernel,
)

try:  # amp decorators as unsloth/kernels/utils.py:45-54
    _custom_fwd = torch.amp.custom_fwd(device_type="cuda")
    _custom_bwd = torch.amp.custom_bwd(device_type="cuda")
except Exception:  # pragma: no cover
    _custom_fwd = lambda f: f
    _custom_bwd = lambda f: f


def _lora_grads(X2d, dY2d, A, B, s, dtype):
    """(dA [r,in], dB [out,r]) for Y = X W^T + s (X A^T) B^T; fast_lora.py:172-189 in the
    un-transposed layout, through the library GEMM (used when the fused products below do not apply).
    Returns (None, None) when the adapter is absent."""
    if A is None:
        return None, None
    At, Bt = cast_lora(A, dtype).t(), cast_lora(B, dtype).t()          # [in,r], [r,out]
    dA_t = torch.empty_like(At)                          # [in, r]
    dB_t = torch.empty_like(Bt)                          # [r, out]
    # d_A = X^T @ (dY @ B^T-as-stored) ; d_B = (A^T-as-stored @ X^T) @ dY     (:172-173)
    dA_t.addmm_(X2d.t(), dY2d @ Bt.t(), alpha=s, beta=0)
    dB_t.addmm_(At.t() @ X2d.t(), dY2d, alpha=s, beta=0)
    return dA_t.t(), dB_t.t()


def _lora_grads_fused(items):
    """items: [(X2d, dY2d, A, B, s, XA, P)] with XA = X @ A^T (fp32, saved by the forward) and P = dY @ B (fp32,
    shared with the dX GEMM). Returns [(dA, dB)] in fp32, all products of the block in ONE launch per 8:
        dA = s * P^T @ X   [r, in]          dB = s * dY^T @ XA   [out, r]          (fast_lora.py:172-189)"""
    probs, slots, targets, sinks = [], [], [], []
    for (X2d, dY2d, A, B, s, XA, P) in items:
        if A is None:
            slots.append(None)
            continue
        r = A.shape[0]
        slots.append(len(probs))
        probs.append((P, X2d, r, False, s))
        probs.append((XA, dY2d, r, True, s))
        for prm in (A, B):
            sink = grad_sink(prm)
            sinks.append((sink, prm))
            targets.append(sink.grad_view(prm) if sink is not None else None)
    outs = lora_tn(probs, targets if any(t is not None for t in targets) else None)
    for sink, prm in sinks:
        if sink is not None:
            sink.ready(prm)               # the gradient is in the arena: autograd gets None for it
    outs = [None if sk[0] is not None else o for o, sk in zip(outs, sinks)]
    return [(None, None) if k is None else (outs[k], outs[k + 1]) for k in slots]


# ---- the blocks as plain functions (forward returns what the backward needs; nothing here touches autograd), used by
#      the autograd.Function wrappers below AND by the whole-layer Function of models/fast_layer.py, which decides
#      per tensor whether to keep it or to recompute it in the backward.
_ACT_NAMES = {swiglu_fg_kernel: "swiglu", swiglu_DWf_DW_dfg_kernel: "swiglu",
              geglu_exact_forward_kernel: "geglu_exact", geglu_exact_backward_kernel: "geglu_exact",
              geglu_approx_forward_kernel: "geglu_approx", geglu_approx_backward_kernel: "geglu_approx"}


MERGE_GATE_UP = True          # module attribute (tests flip it); the environment switch went with its A/B (profiles/r06_gemm_row_padding.jsonl)


def _gate_up(X, gate, up, return_xa=False):
    """(e, g) = X @ W_gate^T, X @ W_up^T (+ LoRA) as the two column halves of ONE row-padded buffer (utils.alloc_rows): the
    in-place activation backward then leaves df | de side by side, so dX = [df | de] @ [W_up; W_gate] is ONE GEMM over the
    concatenated 2 I features (utils._lora_linear_dx_merged -- no read-modify-write of dX, one launch), and every GEMM that
    reads an intermediate as its A operand (down_proj forward: h; that dX: df | de) sees rows that are not a whole number of
    4 KiB pages."""
    outs = None
    widths = [(q.shape[0] if q is not None else W.shape[0]) for (W, q, *_rest) in (gate, up)]
    if (MERGE_GATE_UP and X.is_cuda and X.dtype in (torch.bfloat16, torch.float16) and widths[0] == widths[1]
            and widths[0] % 64 == 0 and all(len(p) <= 5 or p[5] is None for p in (gate, up))):
        M = X.numel() // X.shape[-1]
        eg = alloc_rows(M, 2 * widths[0], X.dtype, X.device)
        outs = [eg[:, :widths[0]], eg[:, widths[0]:]]
    return lora_linear_forward(X, [gate, up], outs=outs, return_xa=return_xa)


def mlp_forward(X, gate, up, down, act_fwd):
    """gate/up/down = (W, W_quant, A, B, s). Returns (out, e, g, (xa_gate, xa_up, xa_down)); fast_lora.py:93-96.
    The activation kernel also produces h @ A_down^T while h is in its registers (utils.glu_fwd_xa) when it can."""
    (e, g), xa_gu = _gate_up(X, gate, up, return_xa=True)
    act = _ACT_NAMES.get(act_fwd)
    e2, g2 = e.view(-1, e.shape[-1]), g.view(-1, g.shape[-1])
    fused = glu_fwd_xa(act, e2, g2, down) if (act is not None and _same_rows([e2, g2])) else None
    if fused is not None:
        h, pre = fused
        (out,), xa_d = lora_linear_forward(h.view(e.shape), [down], return_xa=True, pre_xa=pre)
    else:
        if not e.is_contiguous():                  # (the plain activation kernels take flat buffers)
            e, g = e.contiguous(), g.contiguous()
        h = act_fwd(e, g)
        (out,), xa_d = lora_linear_forward(h, [down], return_xa=True)
    return out, e, g, (xa_gu[0], xa_gu[1], xa_d[0])


def mlp_gate_up_forward(X, gate, up):
    """The recomputable half of mlp_forward: (e, g) only (the backward rebuilds h = act(e) * g itself)."""
    return _gate_up(X, gate, up)


def mlp_backward(dY, X, e, g, xas, gate, up, down, act_bwd, inplace=True):
    """fast_lora.py:127-229. OVERWRITES e and g (the activation backward works in place) and, when `inplace`, writes dX
    into X's buffer. Returns (dX [same shape as X], (d_gateA, d_gateB, d_upA, d_upB, d_downA, d_downB))."""
    xa_g, xa_u, xa_d = xas
    shape = X.shape
    dY = dY.reshape(-1, dY.shape[-1])
    X2 = X.reshape(-1, X.shape[-1])
    e = e.view(-1, e.shape[-1])
    g = g.view(-1, g.shape[-1])
    dtype = X2.dtype
    # DW = dY @ W_down (+ LoRA)                                     fast_lora.py:156
    (p_d,) = lora_dx_terms([dY], [down])
    # (DW shares e's row stride: the activation backward takes ONE `ld` for DW, e and g)
    padded = e.stride(0) != e.shape[1] and _same_rows([e, g])
    DW = lora_linear_dx([dY], [down], terms=[p_d],
                        out=alloc_rows(e.shape[0], e.shape[1], e.dtype, e.device, ld=e.stride(0)) if padded else None)
    act = _ACT_NAMES.get(act_bwd)
    fused = glu_bwd_terms(act, DW, e, g, up, gate) if (act is not None and DW.dim() == 2) else None
    if fused is not None:                                          # activation backward + df @ B_up, de @ B_gate in one pass
        h, df, de, (p_u, p_g) = fused
    else:
        if padded:                                                 # (the plain in-place kernels take flat buffers)
            DW, e, g = DW.contiguous(), e.contiguous(), g.contiguous()
        DW, e, g = act_bwd(DW, e, g)                               # in place (:157)
        h, df, de = DW, e, g
        p_u, p_g = lora_dx_terms([df, de], [up, gate])
    if lora_tn_supported([h, dY, X2, df, de]):
        (d_downA, d_downB), (d_upA, d_upB), (d_gateA, d_gateB) = _lora_grads_fused([
            (h, dY, down[2], down[3], down[4], xa_d, p_d), (X2, df, up[2], up[3], up[4], xa_u, p_u),
            (X2, de, gate[2], gate[3], gate[4], xa_g, p_g)])
    else:
        d_downA, d_downB = _lora_grads(h, dY, down[2], down[3], down[4], dtype)
        d_upA, d_upB = _lora_grads(X2, df, up[2], up[3], up[4], dtype)
        d_gateA, d_gateB = _lora_grads(X2, de, gate[2], gate[3], gate[4], dtype)
    # dX = df @ W_up + de @ W_gate (+ LoRA terms), into X's buffer when inplace (:193-204)
    dX = lora_linear_dx([df, de], [up, gate], out=X2 if (inplace and X2.is_contiguous()) else None, terms=[p_u, p_g])
    return dX.view(shape), (d_gateA, d_gateB, d_upA, d_upB, d_downA, d_downB)


def qkv_forward(X, q, k, v):
    """Returns (Q, K, V, (xa_q, xa_k, xa_v)); fast_lora.py:393-405. Each projection is (W, W_quant, A, B, s) or, with the
    base layer's bias, (W, W_quant, A, B, s, bias): the bias is added in the GEMM epilogue (Qwen2's q/k/v)."""
    (Q, K, V), xa = lora_linear_forward(X, [q, k, v], return_xa=True)
    return Q, K, V, tuple(xa)


def qkv_backward(dQ, dK, dV, X, xas, q, k, v, inplace=True):
    """fast_lora.py:425-540. Returns (dX, (d_QA, d_QB, d_KA, d_KB, d_VA, d_VB)); dX lands in X's buffer when `inplace`."""
    xa_q, xa_k, xa_v = xas
    shape = X.shape
    dQ = dQ.reshape(-1, dQ.shape[-1])
    dK = dK.reshape(-1, dK.shape[-1])
    dV = dV.reshape(-1, dV.shape[-1])
    X2 = X.reshape(-1, X.shape[-1])
    projs = [q, k, v]
    p_q, p_k, p_v = lora_dx_terms([dQ, dK, dV], projs)
    if lora_tn_supported([X2, dQ, dK, dV]):
        (d_QA, d_QB), (d_KA, d_KB), (d_VA, d_VB) = _lora_grads_fused([
            (X2, dQ, q[2], q[3], q[4], xa_q, p_q), (X2, dK, k[2], k[3], k[4], xa_k, p_k),
            (X2, dV, v[2], v[3], v[4], xa_v, p_v)])
    else:
        d_QA, d_QB = _lora_grads(X2, dQ, q[2], q[3], q[4], X2.dtype)
        d_KA, d_KB = _lora_grads(X2, dK, k[2], k[3], k[4], X2.dtype)
        d_VA, d_VB = _lora_grads(X2, dV, v[2], v[3], v[4], X2.dtype)
    # dX accumulated over q, k, v; overwrites X when inplace (fast_lora.py:497-517)
    dX = lora_linear_dx([dQ, dK, dV], projs, out=X2 if (inplace and X2.is_contiguous()) else None,
                        terms=[p_q, p_k, p_v])
    return dX.view(shape), (d_QA, d_QB, d_KA, d_KB, d_VA, d_VB)


def w_forward(X, proj):
    """Returns (out, xa); fast_lora.py:600-611."""
    (XW,), xa = lora_linear_forward(X, [proj], return_xa=True)
    return XW, xa[0]


def w_backward(dY, X, xa, proj):
    """fast_lora.py:613-650. Returns (dX, (d_A, d_B))."""
    shape = X.shape
    dY = dY.reshape(-1, dY.shape[-1])
    X2 = X.reshape(-1, X.shape[-1])
    (p,) = lora_dx_terms([dY], [proj])
    if lora_tn_supported([X2, dY]):
        ((d_A, d_B),) = _lora_grads_fused([(X2, dY, proj[2], proj[3], proj[4], xa, p)])
    else:
        d_A, d_B = _lora_grads(X2, dY, proj[2], proj[3], proj[4], X2.dtype)
    dX = lora_linear_dx([dY], [proj], terms=[p])
    return dX.view(shape), (d_A, d_B)


class LoRA_MLP(torch.autograd.Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, X, gateW, gateW_quant, gateA, gateB, gateS, upW, upW_quant, upA, upB, upS,
                downW, downW_quant, downA, downB, downS, _forward_function, _backward_function,
                inplace=True):
        i, e, g, xas = mlp_forward(X, (gateW, gateW_quant, gateA, gateB, gateS), (upW, upW_quant, upA, upB, upS),
                                   (downW, downW_quant, downA, downB, downS), _forward_function)
        ctx.custom_saved_tensors = (gateW, gateW_quant, gateS, upW, upW_quant, upS, downW,
                                    downW_quant, downS, _backward_function)
        ctx.save_for_backward(gateA, gateB, upA, upB, downA, downB, X, e, g)
        ctx.xa = xas                                # X A_g^T, X A_u^T, h A_d^T (fp32 [M, r]): tiny, no grad
        ctx.inplace = inplace
        return i

    @staticmethod
    @_custom_bwd
    def backward(ctx, dY):
        (gateW, gateW_quant, gateS, upW, upW_quant, upS, downW, downW_quant, downS,
         _backward_function) = ctx.custom_saved_tensors
        gateA, gateB, upA, upB, downA, downB, X, e, g = ctx.saved_tensors
        dX, (d_gateA, d_gateB, d_upA, d_upB, d_downA, d_downB) = mlp_backward(
            dY, X, e, g, ctx.xa, (gateW, gateW_quant, gateA, gateB, gateS), (upW, upW_quant, upA, upB, upS),
            (downW, downW_quant, downA, downB, downS), _backward_function, ctx.inplace)
        return (dX, None, None, d_gateA, d_gateB, None, None, None, d_upA, d_upB, None,
                None, None, d_downA, d_downB, None, None, None, None)


def _apply_mlp(self, X, fwd, bwd, inplace=True):
    gateW, gateW_quant, gateA, gateB, gateS = get_lora_parameters(self.gate_proj)
    upW, upW_quant, upA, upB, upS = get_lora_parameters(self.up_proj)
    downW, downW_quant, downA, downB, downS = get_lora_parameters(self.down_proj)
    return LoRA_MLP.apply(X, gateW, gateW_quant, gateA, gateB, gateS, upW, upW_quant, upA, upB, upS,
                          downW, downW_quant, downA, downB, downS, fwd, bwd, inplace)


def apply_lora_mlp_swiglu(self, X, inplace=True):
    return _apply_mlp(self, X, swiglu_fg_kernel, swiglu_DWf_DW_dfg_kernel, inplace)


def apply_lora_mlp_geglu_exact(self, X, inplace=True):
    return _apply_mlp(self, X, geglu_exact_forward_kernel, geglu_exact_backward_kernel, inplace)


def apply_lora_mlp_geglu_approx(self, X):
    return _apply_mlp(self, X, geglu_approx_forward_kernel, geglu_approx_backward_kernel)


def _bias_grad(dY, bias, needed):
    """d bias = column sums of dY (only when the bias itself trains: full fine-tuning / bias="all")."""
    if bias is None or not needed:
        return None
    return dY.reshape(-1, dY.shape[-1]).sum(dim=0, dtype=torch.float32).to(bias.dtype)


class LoRA_QKV(torch.autograd.Function):
    """The reference's signature (fast_lora.py:335-540) plus three optional trailing arguments: the base layers' biases.
    The reference refuses biased projections on this path (llama.py:3695-3772: Qwen2 falls back to PEFT's forward);
    here the bias rides in the GEMM epilogue and, being an additive constant, drops out of dX and of every LoRA gradient."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, X, QW, QW_quant, QA, QB, QS, KW, KW_quant, KA, KB, KS, VW, VW_quant, VA, VB, VS,
                inplace=True, Qb=None, Kb=None, Vb=None):
        Q, K, V, xa = qkv_forward(X, (QW, QW_quant, QA, QB, QS, Qb), (KW, KW_quant, KA, KB, KS, Kb),
                                  (VW, VW_quant, VA, VB, VS, Vb))
        ctx.custom_saved_tensors = (QW, QW_quant, QS, KW, KW_quant, KS, VW, VW_quant, VS, Qb, Kb, Vb)
        ctx.save_for_backward(X, QA, QB, KA, KB, VA, VB)
        ctx.xa = xa
        ctx.inplace = inplace
        return Q, K, V

    @staticmethod
    @_custom_bwd
    def backward(ctx, dQ, dK, dV):
        QW, QW_quant, QS, KW, KW_quant, KS, VW, VW_quant, VS, Qb, Kb, Vb = ctx.custom_saved_tensors
        X, QA, QB, KA, KB, VA, VB = ctx.saved_tensors
        nig = tuple(ctx.needs_input_grad) + (False,) * 3          # (the bias arguments are optional)
        d_bias = (_bias_grad(dQ, Qb, nig[17]), _bias_grad(dK, Kb, nig[18]), _bias_grad(dV, Vb, nig[19]))
        dX, (d_QA, d_QB, d_KA, d_KB, d_VA, d_VB) = qkv_backward(
            dQ, dK, dV, X, ctx.xa, (QW, QW_quant, QA, QB, QS), (KW, KW_quant, KA, KB, KS), (VW, VW_quant, VA, VB, VS),
            ctx.inplace)
        grads = (dX, None, None, d_QA, d_QB, None, None, None, d_KA, d_KB, None, None, None, d_VA, d_VB, None, None)
        return grads + d_bias if len(ctx.needs_input_grad) > 17 else grads


def apply_lora_qkv(self, X, inplace=True):
    QW, QW_quant, QA, QB, QS = get_lora_parameters(self.q_proj)
    KW, KW_quant, KA, KB, KS = get_lora_parameters(self.k_proj)
    VW, VW_quant, VA, VB, VS = get_lora_parameters(self.v_proj)
    biases = tuple(getattr(getattr(p, "base_layer", p), "bias", None) for p in (self.q_proj, self.k_proj, self.v_proj))
    if all(b is None for b in biases):
        return LoRA_QKV.apply(X, QW, QW_quant, QA, QB, QS, KW, KW_quant, KA, KB, KS, VW, VW_quant, VA, VB,
                              VS, inplace)
    return LoRA_QKV.apply(X, QW, QW_quant, QA, QB, QS, KW, KW_quant, KA, KB, KS, VW, VW_quant, VA, VB,
                          VS, inplace, *biases)


class LoRA_W(torch.autograd.Function):
    @staticmethod
    @_custom_fwd
    def forward(ctx, X, W, W_quant, A, B, S, bias=None):
        XW, xa = w_forward(X, (W, W_quant, A, B, S, bias))
        ctx.custom_saved_tensors = (W, W_quant, S, bias)
        ctx.save_for_backward(A, B, X)
        ctx.xa = xa
        return XW

    @staticmethod
    @_custom_bwd
    def backward(ctx, dY):
        W, W_quant, S, bias = ctx.custom_saved_tensors
        A, B, X = ctx.saved_tensors
        d_bias = _bias_grad(dY, bias, len(ctx.needs_input_grad) > 6 and ctx.needs_input_grad[6])
        dX, (d_A, d_B) = w_backward(dY, X, ctx.xa, (W, W_quant, A, B, S))
        return (dX, None, None, d_A, d_B, None, d_bias) if len(ctx.needs_input_grad) > 6 else (dX, None, None, d_A, d_B, None)


def apply_lora_o(self, X):
    OW, OW_quant, OA, OB, OS = get_lora_parameters(self.o_proj)
    bias = getattr(getattr(self.o_proj, "base_layer", self.o_proj), "bias", None)
    if bias is None:
        return LoRA_W.apply(X, OW, OW_quant, OA, OB, OS)
    return LoRA_W.apply(X, OW, OW_quant, OA, OB, OS, bias)


@torch._disable_dynamo
def fast_lora_forward(self, x, *args, **kwargs):
    raise NotImplementedError("Unsloth: Currently not supported yet - reshaping done incorrectly")
