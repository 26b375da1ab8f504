"""Per-token log-probabilities for GRPO / DPO without a [B, L, V] logits tensor.

The reference imports `chunked_hidden_states_selective_log_softmax` and `chunked_selective_log_softmax` from
unsloth_zoo (unsloth/models/rl_replacements.py:29-37; call sites :1989-2006, :2255-2272): third-party code that is
not in the repository, so the semantics below are restated from the call sites and pinned only against the plain
formula `log_softmax(f(h @ W^T))[index]` (parity unpinned, like unsloth_fused_ce_loss):
    logits = hidden @ lm_head^T
    logits *= logit_scale_multiply (if != 0);  logits /= logit_scale_divide (if != 0)
    logits  = cap * tanh(logits / cap)         (if logit_softcapping != 0)
    logits /= temperature                      (if != 1)
    out[b, l] = logits[b, l, index[b, l]] - logsumexp(logits[b, l, :])            (fp32)
Built from the same pieces as the fused linear cross-entropy: row chunks, MFMA GEMM into a transient chunk of
logits, the single-pass CE kernel (log-prob = -loss), d(hidden) computed in the forward because lm_head is frozen
and a log-prob depends on its own row only (the backward is a row-wise scale)."""
import torch

from .. import _lib
from ..kernels import utils as _u
from ..kernels.cross_entropy_loss import (Fast_CrossEntropyLoss, _ce_backward_, _ce_forward, _dhidden, _logits_chunk,
                                          _nn_ok, _transposed_weight)


class _ChunkedLogProbs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden2d, weight, weight_t, index, softcap, scale, chunk_rows):
        T, H = hidden2d.shape
        V = weight.shape[0]
        dev = hidden2d.device
        out = torch.empty(T, dtype=torch.float32, device=dev)
        need_grad = hidden2d.requires_grad
        dh = torch.empty_like(hidden2d) if need_grad else None
        for r0 in range(0, T, chunk_rows):
            r1 = min(T, r0 + chunk_rows)
            chunk = _logits_chunk(r1 - r0, V, hidden2d.dtype, dev)       # row stride padded to a multiple of 8
            logits = chunk[:, :V]
            _u._launch_gemm(hidden2d[r0:r1], [_u._group(weight, logits, V, weight.stride(0))], nf4=False)
            idx = index[r0:r1]
            losses, lse = _ce_forward(logits, idx, softcap, scale)
            out[r0:r1] = -losses
            if need_grad:
                dl = torch.full((r1 - r0,), -1.0, dtype=torch.float32, device=dev)      # d(logprob) = -d(loss)
                _ce_backward_(logits, dl, lse, idx, softcap, scale)                     # logits <- d logprob / d logits
                _dhidden(chunk, logits, weight, weight_t, dh[r0:r1])
        ctx.save_for_backward(dh)
        return out

    @staticmethod
    def backward(ctx, g):
        (dh,) = ctx.saved_tensors
        if dh is None:
            return None, None, None, None, None, None, None
        # row-wise upstream scale applied in fp32, one rounding (not g rounded to bf16 first)
        return (dh.to(torch.float32) * g.to(torch.float32).unsqueeze(1)).to(dh.dtype), None, None, None, None, None, None


def _effective_scale(mult, div, softcap, temperature):
    s = 1.0
    if mult:
        s *= float(mult)
    if div:
        s /= float(div)
    if temperature not in (None, 0, 1, 1.0):
        if softcap:
            raise NotImplementedError("logit_softcapping together with temperature != 1: the CE kernel scales before "
                                      "the soft cap (cross_entropy_loss.py:35-120), not after")
        s /= float(temperature)
    return 0.0 if s == 1.0 else s


def chunked_hidden_states_selective_log_softmax(hidden_states, lm_head, index, chunks=4, logit_scale_multiply=0.0,
                                                logit_scale_divide=0.0, logit_softcapping=0.0, temperature=1.0):
    """hidden_states [B, L, H] (activation dtype), lm_head [V, H] frozen, index [B, L] int -> log-probs [B, L] fp32."""
    _lib.require_gpu(hidden_states, lm_head, index)
    if lm_head.requires_grad:
        raise NotImplementedError("frozen lm_head only (LoRA fine-tuning keeps it frozen)")
    B, L, H = hidden_states.shape
    h2d = hidden_states.reshape(-1, H)
    if h2d.stride(1) != 1 or h2d.stride(0) % 8:
        h2d = h2d.contiguous()
    W = lm_head.detach()
    if W.dtype != h2d.dtype:
        W = W.to(h2d.dtype)
    T = B * L
    chunks = max(1, int(chunks))
    chunk_rows = min(4096, max(256, -(-T // chunks)))         # <= 1 GB of transient logits at vocab 128k
    nn = _nn_ok(min(chunk_rows, T), W.shape[0], H) and W.stride(1) == 1 and W.stride(0) % 8 == 0
    Wt = None if nn else _transposed_weight(W)
    scale = _effective_scale(logit_scale_multiply, logit_scale_divide, logit_softcapping, temperature)
    out = _ChunkedLogProbs.apply(h2d, W, Wt, index.reshape(-1).to(torch.int64).contiguous(),
                                 float(logit_softcapping or 0), scale, int(chunk_rows))
    return out.view(B, L)


def chunked_selective_log_softmax(logits, index, temperature=1.0, chunks=4):
    """logits [B, L, V] already produced by the model (scaling / soft cap applied there) -> log-probs [B, L] fp32.
    The CE backward writes d(logits) in place over `logits`, like the reference's Fast_CrossEntropyLoss."""
    _lib.require_gpu(logits, index)
    B, L, V = logits.shape
    scale = _effective_scale(0, 0, 0, temperature)
    losses = Fast_CrossEntropyLoss.apply(logits.reshape(B * L, V), index.reshape(-1).to(torch.int64), 0, scale)
    return (-losses).view(B, L)
