"""Cross-entropy through the HIP kernels; mirror of unsloth/kernels/cross_entropy_loss.py, plus
the fused linear + cross-entropy entry point the reference imports from unsloth_zoo
(`unsloth_fused_ce_loss`, call site unsloth/models/llama.py:1497-1509).

  Fast_CrossEntropyLoss     (:288-418)  per-row loss, saves (logits, logsumexp, labels), backward
                                        overwrites logits with the gradient and returns that buffer
  fast_cross_entropy_loss   (:421-449)  sum / n_items
  patch_loss_functions      (:459-473)  installs the fast loss into transformers' LOSS_MAPPING

Vocabulary size is unlimited here (one streaming launch), so the reference's split into a
<=65536 kernel and a chunked kernel + host torch.logsumexp (:303-370) disappears; results are
the same quantity (logsumexp over the whole row).
"""
import torch

from .. import _lib
from . import utils as _u


def _ce_forward(logits2d, labels, softcap, scale):
    n_rows, vocab = logits2d.shape
    losses = torch.empty(n_rows, dtype=torch.float32, device=logits2d.device)
    lse = torch.empty(n_rows, dtype=torch.float32, device=logits2d.device)
    with _lib.device_ctx(logits2d):
        rc = _lib.lib().uamd_cross_entropy_forward(
            _lib.ptr(logits2d), logits2d.stride(0), _lib.ptr(losses), _lib.ptr(lse), _lib.ptr(labels),
            n_rows, vocab, float(softcap), float(scale), _lib.dtype_code(logits2d.dtype),
            _lib.stream_of(logits2d))
    _lib.check(rc, "uamd_cross_entropy_forward")
    return losses, lse


def _ce_backward_(logits2d, dlosses, lse, labels, softcap, scale):
    n_rows, vocab = logits2d.shape
    with _lib.device_ctx(logits2d):
        rc = _lib.lib().uamd_cross_entropy_backward(
            _lib.ptr(logits2d), logits2d.stride(0), _lib.ptr(dlosses), dlosses.stride(0), _lib.ptr(lse),
            _lib.ptr(labels), n_rows, vocab, float(softcap), float(scale),
            _lib.dtype_code(logits2d.dtype), _lib.stream_of(logits2d))
    _lib.check(rc, "uamd_cross_entropy_backward")
    return logits2d


class Fast_CrossEntropyLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, logit_softcapping=0, logit_scaling=0):
        _lib.require_gpu(logits)
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        losses, lse = _ce_forward(logits, labels, logit_softcapping or 0, logit_scaling or 0)
        ctx.save_for_backward(logits, lse, labels)
        ctx.logit_softcapping = logit_softcapping or 0
        ctx.logit_scaling = logit_scaling or 0
        return losses

    @staticmethod
    def backward(ctx, dlosses):
        logits, lse, labels = ctx.saved_tensors
        dlosses = dlosses.to(torch.float32)
        if dlosses.dim() == 0 or dlosses.stride(0) == 0:
            dlosses = dlosses.expand(logits.shape[0]).contiguous()
        # in place over the saved logits, returned as the gradient (:413-418)
        _ce_backward_(logits, dlosses, lse, labels, ctx.logit_softcapping, ctx.logit_scaling)
        return logits, None, None, None


def fast_cross_entropy_loss(logits, labels, logit_softcapping=0, logit_scaling=0, n_items=None):
    """logits (batch, seq_len, vocab), labels (batch, seq_len) -> scalar; :421-449."""
    batch, seq_len, d = logits.shape
    assert labels.shape == (batch, seq_len)
    loss = Fast_CrossEntropyLoss.apply(
        logits.view(batch * seq_len, d), labels.view(-1), logit_softcapping, logit_scaling)
    if n_items is None:
        n_items = torch.count_nonzero(labels != -100)
    if torch.is_tensor(n_items):
        n_items = n_items.to(logits.device)
    return loss.sum() / n_items


def _unsloth_causal_lm_loss(logits, labels, vocab_size=None, num_items_in_batch=None,
                            ignore_index=-100, **kwargs):
    """ForCausalLM loss with the fast kernel: shift labels (llama.py:1545-1551) then CE."""
    shift_labels = kwargs.get("shift_labels")
    if shift_labels is None:
        shift_labels = torch.empty_like(labels)
        shift_labels[..., :-1] = labels[..., 1:]
        shift_labels[..., -1] = -100
    if logits.dim() == 2:
        logits = logits.unsqueeze(0)
    return fast_cross_entropy_loss(logits, shift_labels.view(logits.shape[0], -1), n_items=num_items_in_batch)


def post_patch_loss_function(model):
    """unsloth_zoo.loss_utils.post_patch_loss_function: point the instance at the patched loss."""
    try:
        model.loss_function = _unsloth_causal_lm_loss
    except Exception:
        pass
    return model


_LOSS_ORIGINALS = {}


def patch_loss_functions(torch_compile=True):
    """:459-473: route transformers' ForCausalLM loss (and aliases of it) to the fast kernel."""
    try:
        import transformers.loss.loss_utils as _lu
        for key, fn in list(_lu.LOSS_MAPPING.items()):
            if key == "ForCausalLM" or getattr(fn, "__name__", "") == "ForCausalLMLoss":
                _LOSS_ORIGINALS.setdefault(key, fn)
                _lu.LOSS_MAPPING[key] = _unsloth_causal_lm_loss
    except (ImportError, AttributeError):
        pass


def unpatch_loss_functions():
    """Undo patch_loss_functions()."""
    try:
        import transformers.loss.loss_utils as _lu
        for key, fn in _LOSS_ORIGINALS.items():
            _lu.LOSS_MAPPING[key] = fn
        _LOSS_ORIGINALS.clear()
    except (ImportError, AttributeError):
        pass


# ------------------------------------------------------------------------------------------------
def _padded_vocab(V):
    return (V + 7) // 8 * 8


def _logits_chunk(rows, V, dtype, device):
    """[rows, Vp] buffer, Vp = V rounded up to a multiple of 8 (the MFMA GEMM wants 16-byte aligned rows and a
    contraction length that is a multiple of 8: a vocabulary like 32001 -- one added pad token -- must not make
    every training step raise). The GEMM and the CE kernels work on the [:, :V] view; the padding columns are zero
    and stay zero, so the full view is a valid A operand for d(hidden) = dlogits @ W (with W^T zero-padded alike)."""
    Vp = _padded_vocab(V)
    buf = torch.empty((rows, Vp), dtype=dtype, device=device)
    if Vp != V:
        buf[:, V:].zero_()
    return buf


def fused_ce_chunk_rows(T, V, itemsize, device, target_gb=None):
    """Rows of hidden states per logits chunk. The reference (unsloth_zoo.loss_utils, called with `target_gb` from
    models/llama.py:1497-1509) sizes its chunks from the free VRAM / an explicit `target_gb`; same policy here:
    the transient [rows, V] logits chunk takes at most `target_gb` GiB when given, else at most 1/8 of the memory
    that is free right now, and never more than 4096 rows (1.05 GB at vocab 128256 in bf16: beyond that the GEMM
    gains nothing). Multiples of 256 rows (one GEMM tile), at least 256."""
    per_row = _padded_vocab(V) * itemsize
    if target_gb is not None and target_gb > 0:
        budget = float(target_gb) * (1 << 30)
    else:
        try:
            free, _ = torch.cuda.mem_get_info(device)
        except Exception:
            free = 8 << 30
        budget = free / 8
    rows = int(budget // per_row) // 256 * 256
    return max(256, min(4096, rows, (T + 255) // 256 * 256))


def _nn_ok(rows, V, H):
    """d(hidden) = dlogits [rows, V] @ W [V, H] can contract over W's ROWS in place (NN form of the 256-tile GEMM):
    no transposed copy of the lm_head (1.05 GB at Llama-3's vocabulary) has to exist."""
    return _u.NN_DX and V % 64 == 0 and H % 8 == 0 and _u._use_gemm256(rows, V, [H])


def _dhidden(chunk, dlogits, weight, weight_t, out):
    """out = dlogits @ W. `chunk` is the zero-padded [rows, Vp] buffer `dlogits` is a view of; `weight_t` (W^T,
    [H, Vp]) is only built when the NN form does not apply."""
    V, H = weight.shape
    if weight_t is None:
        _u._launch_gemm(dlogits, [_u._group(weight, out, H, weight.stride(0))], nf4=False, nn=True)
    else:
        _u._launch_gemm(chunk, [_u._group(weight_t, out, H, weight_t.stride(0))], nf4=False)


_DW_SCALE_ROWS = 8192        # rows of the lm_head gradient scaled in fp32 at a time (64-128 MB instead of 2 GB)


class _FusedLinearCE(torch.autograd.Function):
    """loss = sum_rows CE(hidden @ W^T) / n_items without ever holding [T, V] logits.

    Semantics of unsloth_zoo's unsloth_fused_ce_loss (third party, parity unpinned):
    labels are already shifted by the caller of this Function; rows are processed in chunks; per
    chunk: logits = h @ W^T in the activation dtype (MFMA GEMM), CE in fp32 inside the kernel,
    d(hidden) computed in the forward (W is frozen) and scaled by the upstream scalar in backward.
    """

    @staticmethod
    def forward(ctx, hidden2d, weight, weight_t, labels, n_items, softcap, scale, chunk_rows, weight_param=None, grad_on=True):
        """`weight_param`: the lm_head Parameter when it TRAINS (full fine-tuning; `weight` is its detached value): its
        gradient dW = sum over chunks dlogits^T @ h is accumulated chunk by chunk (uamd_gemm_tn_256) -- like d(hidden),
        inside the forward, while the chunk of dlogits exists."""
        T, H = hidden2d.shape
        V = weight.shape[0]
        dev = hidden2d.device
        loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        # `grad_on`: torch.is_grad_enabled() of the CALLER (inside Function.forward grad mode is always off): an evaluation
        # pass under no_grad() must not pay the CE backward, the d(hidden) GEMM and the chunked dW GEMMs (3x the compute and a
        # [T, H] + [Vp, H] pair of buffers for a full fine-tuning model)
        train_w = grad_on and weight_param is not None and weight_param.requires_grad
        need_grad = grad_on and (hidden2d.requires_grad or train_w)
        dh = torch.empty_like(hidden2d) if need_grad else None
        dW = None
        inv_n = (1.0 / n_items) if not torch.is_tensor(n_items) else (1.0 / n_items.to(torch.float32))
        for r0 in range(0, T, chunk_rows):
            r1 = min(T, r0 + chunk_rows)
            h = hidden2d[r0:r1]
            chunk = _logits_chunk(r1 - r0, V, hidden2d.dtype, dev)
            logits = chunk[:, :V]
            _u._launch_gemm(h, [_u._group(weight, logits, V, weight.stride(0))], nf4=False)
            lab = labels[r0:r1]
            losses, lse = _ce_forward(logits, lab, softcap, scale)
            loss_sum += losses.sum()
            if need_grad:
                dl = torch.ones(r1 - r0, dtype=torch.float32, device=dev) * inv_n
                _ce_backward_(logits, dl, lse, lab, softcap, scale)        # logits <- dlogits
                _dhidden(chunk, logits, weight, weight_t, dh[r0:r1])
                if train_w:
                    # the padded chunk (zero columns beyond V) keeps the GEMM's 8-column granularity for any vocabulary
                    if dW is None:
                        dW = torch.empty((chunk.shape[1], H), dtype=hidden2d.dtype, device=dev)
                    _u.dense_dw(chunk, h, out=dW, accumulate=r0 > 0)
        ctx.save_for_backward(dh, dW)
        ctx.weight_param = weight_param if train_w else None
        return loss_sum * inv_n

    @staticmethod
    def backward(ctx, dloss):
        dh, dW = ctx.saved_tensors
        scale = dloss.to(torch.float32)
        d_weight = None
        if dW is not None:
            P = ctx.weight_param
            V = P.shape[0]
            sink = _u.grad_sink(P)
            # the upstream scale in fp32, ONE rounding (like d(hidden) below): the scalar rounded to bf16 first would bias the
            # lm_head gradient by up to 2^-9 against every other parameter's. In ROW CHUNKS: a whole [V, H] fp32 temporary is
            # 2.1 GB for a 128k x 4096 head, at the loss peak of a full fine-tuning step
            if sink is not None:
                view = sink.grad_view(P)
                first = sink.first_write(P)
            else:
                view, first = torch.empty((V, dW.shape[1]), dtype=P.dtype, device=dW.device), True
            for r0 in range(0, V, _DW_SCALE_ROWS):
                r1 = min(V, r0 + _DW_SCALE_ROWS)
                part = dW[r0:r1].to(torch.float32) * scale
                if first:
                    view[r0:r1].copy_(part)
                else:
                    view[r0:r1].add_(part.to(view.dtype))
            if sink is not None:
                sink.ready(P)
            else:
                d_weight = view
        if dh is not None and ctx.needs_input_grad[0]:
            # the upstream scale (1/accumulation steps, a GradScaler factor, ...) is applied in fp32 and the
            # product rounded once: casting the scalar to bf16 first would put 2^-9 of relative error on every
            # gradient of the step
            dh = (dh.to(torch.float32) * scale).to(dh.dtype)
        else:
            dh = None
        return (dh, None, None, None, None, None, None, None, d_weight, None)[:len(ctx.needs_input_grad)]


def _transposed_weight(weight, owner=None):
    """W^T [H, Vp] of the frozen lm_head (vocabulary zero-padded to a multiple of 8) for the shapes the NN form of the
    GEMM does not take. Cached ON `owner` (the lm_head Parameter: the copy lives and dies with it, and is rebuilt when
    its version or storage changes); without an owner it is built per call -- a cache keyed on the data pointer alone
    would hand a NEW tensor that the allocator placed at a freed lm_head's address the old model's transpose."""
    V, H = weight.shape
    Vp = _padded_vocab(V)
    if owner is not None:
        ent = getattr(owner, "_uamd_wt", None)
        if ent is not None and ent[0] == owner._version and ent[1] == weight.data_ptr() and ent[2].shape == (H, Vp) \
                and ent[2].dtype == weight.dtype:
            return ent[2]
    if Vp == V:
        wt = weight.detach().t().contiguous()
    else:
        wt = torch.zeros((H, Vp), dtype=weight.dtype, device=weight.device)
        wt[:, :V] = weight.detach().t()
    if owner is not None:
        owner._uamd_wt = (owner._version, weight.data_ptr(), wt)
    return wt


def unsloth_fused_ce_loss(trainer, hidden_states, lm_head_weight, lm_head_bias, labels, mask=None,
                          n_items=None, scaling=None, target_gb=None, torch_compile=True,
                          logit_softcapping=0, logit_scaling=0, chunk_rows=None, **kwargs):
    """Same call signature as the reference's import (llama.py:1497-1509): shifts labels
    internally, never materialises [T, V], returns the scalar mean loss over n_items."""
    if lm_head_bias is not None:
        raise NotImplementedError("fused linear-CE: bias-free lm_head only")
    _lib.require_gpu(hidden_states)
    H = hidden_states.shape[-1]
    shift = torch.empty_like(labels)
    shift[..., :-1] = labels[..., 1:]
    shift[..., -1] = -100
    if mask is not None:
        shift = torch.where(mask.bool(), shift, torch.full_like(shift, -100))
    shift = shift.reshape(-1).to(torch.int64).contiguous()
    if n_items is None:
        n_items = torch.count_nonzero(shift != -100)
    h2d = hidden_states.reshape(-1, H)
    if h2d.stride(1) != 1 or h2d.stride(0) % 8:
        h2d = h2d.contiguous()
    W = lm_head_weight.detach()
    if W.dtype != h2d.dtype:
        W = W.to(h2d.dtype)
    if chunk_rows is None:
        chunk_rows = fused_ce_chunk_rows(h2d.shape[0], W.shape[0], h2d.element_size(), h2d.device, target_gb)
    nn = _nn_ok(min(int(chunk_rows), h2d.shape[0]), W.shape[0], W.shape[1]) and W.stride(1) == 1 and W.stride(0) % 8 == 0
    Wt = None if nn else _transposed_weight(W, lm_head_weight if W.dtype == lm_head_weight.dtype else None)
    if lm_head_weight.requires_grad:
        if W.dtype != lm_head_weight.dtype:
            raise NotImplementedError("fused linear-CE with a trainable lm_head: weight and activations in one dtype")
        loss = _FusedLinearCE.apply(h2d, W, Wt, shift, n_items, logit_softcapping or 0, logit_scaling or 0,
                                    int(chunk_rows), lm_head_weight if torch.is_grad_enabled() else None, torch.is_grad_enabled())
    else:
        loss = _FusedLinearCE.apply(h2d, W, Wt, shift, n_items, logit_softcapping or 0, logit_scaling or 0,
                                    int(chunk_rows), None, torch.is_grad_enabled())
    if scaling is not None:
        loss = loss * scaling
    return loss
