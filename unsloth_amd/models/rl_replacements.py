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
    Wt = None if nn else _transposed_weight(W, lm_head if W.dtype == lm_head.dtype else None)
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


# ======================================================================================================================
# GRPO / DPO drivers over the chunked log-prob kernels.
#
# Reference call sites: unsloth/models/rl_replacements.py:1466-1700 (`_get_per_token_logps_and_entropies`: left-pack the
# padding, ONE packed varlen forward with UNSLOTH_RETURN_HIDDEN_STATES=1, lm_head on the completion positions only through
# `chunked_hidden_states_selective_log_softmax`, scatter back), :2622-3005 (`compute_loss`: hands the log-probs to
# unsloth_zoo's `grpo_compute_loss_slow` / `grpo_accumulated_loss`). The zoo functions are third party and absent, so
# the loss below restates the PUBLISHED algorithm they implement -- TRL's GRPOTrainer._compute_loss (GRPO / BNPO /
# Dr. GRPO / DAPO aggregation, two-sided clipping, optional delta clip, k3 KL estimator) -- and DPO's sigmoid loss;
# parity is pinned against plain torch restatements of those formulas on an fp32 HF model (tests/test_gpu_rl_drivers.py).
import os as _os


def _packed_completion_index(input_ids, attention_mask, logits_to_keep):
    """Index bookkeeping of the packed forward (integer work). Rows are [left-padded prompt | right-padded completion];
    `logits_to_keep` = the completion columns. Returns
      flat_ids [1, N] the non-padding tokens of all rows back to back, pos [1, N] positions restarting per row,
      lens int32 [rows with tokens], src [n] flat indices whose NEXT token is a completion token of the same row,
      tgt_ids [n] those next tokens, dst (row [n], col [n]) where their log-prob goes in the [B, logits_to_keep] result."""
    B, T = input_ids.shape
    keep = attention_mask.bool() if attention_mask is not None else torch.ones_like(input_ids, dtype=torch.bool)
    lens = keep.sum(dim=1)
    nz = keep.nonzero(as_tuple=False)                         # [N, 2] (row, col), row-major
    flat_ids = input_ids[keep].unsqueeze(0)
    pos = (keep.cumsum(dim=1) - 1)[keep].unsqueeze(0)
    same_row = nz[1:, 0] == nz[:-1, 0]
    is_completion = nz[1:, 1] >= T - logits_to_keep
    sel = same_row & is_completion
    src = sel.nonzero(as_tuple=False).squeeze(1)              # flat index t: hidden[t] predicts flat token t + 1
    tgt = nz[1:][sel]
    return flat_ids, pos, lens[lens > 0].to(torch.int32), src, flat_ids[0, 1:][sel], (tgt[:, 0], tgt[:, 1] - (T - logits_to_keep))


def get_per_token_logps_and_entropies(model, input_ids, attention_mask, logits_to_keep, temperature=1.0, chunks=4,
                                      compute_entropy=False):
    """log p(token | prefix) of the last `logits_to_keep` columns: [B, logits_to_keep] fp32, 0 at padding. ONE packed
    forward over the non-padding tokens (block-diagonal causal attention through the band kernels, position ids restarting
    per row -- no left-pad RoPE error), hidden states instead of logits, lm_head + log-softmax only on the completion
    positions, in row chunks. Differentiable w.r.t. the model's trainable parameters."""
    flat_ids, pos, lens, src, tgt_ids, (dst_r, dst_c) = _packed_completion_index(input_ids, attention_mask, logits_to_keep)
    base = model.get_base_model() if hasattr(model, "get_base_model") else model
    cfg = base.config
    prev = _os.environ.get("UNSLOTH_RETURN_HIDDEN_STATES")
    _os.environ["UNSLOTH_RETURN_HIDDEN_STATES"] = "1"
    try:
        hidden = model(input_ids=flat_ids, position_ids=pos.to(torch.int32), packed_seq_lengths=lens,
                       use_cache=False).logits                  # hidden states in the logits slot (llama.py:1448-1457)
    finally:
        if prev is None:
            _os.environ.pop("UNSLOTH_RETURN_HIDDEN_STATES", None)
        else:
            _os.environ["UNSLOTH_RETURN_HIDDEN_STATES"] = prev
    lm_head = base.get_output_embeddings().weight
    rows = hidden[0].index_select(0, src).unsqueeze(0)          # [1, n, H]
    mult = getattr(cfg, "logit_scale", 0) or 0
    div = getattr(cfg, "logits_scaling", 0) or 0
    cap = getattr(cfg, "final_logit_softcapping", 0) or 0
    B = input_ids.shape[0]
    out = torch.zeros(B, logits_to_keep, dtype=torch.float32, device=input_ids.device)
    ent = None
    if rows.shape[1]:
        lp = chunked_hidden_states_selective_log_softmax(rows, lm_head, tgt_ids.unsqueeze(0), chunks, mult, div, cap,
                                                         temperature)[0]
        out = out.index_put((dst_r, dst_c), lp)
        if compute_entropy:
            # (diagnostic path, no gradient: plain torch over row chunks -- the kernels keep no per-row entropy)
            ent = torch.zeros_like(out)
            with torch.no_grad():
                W = lm_head.float()
                for r0 in range(0, rows.shape[1], 2048):
                    lg = rows[0, r0:r0 + 2048].float() @ W.t()
                    if mult:
                        lg = lg * mult
                    if div:
                        lg = lg / div
                    if cap:
                        lg = cap * torch.tanh(lg / cap)
                    lg = lg / temperature
                    p = torch.softmax(lg, dim=-1)
                    e = torch.logsumexp(lg, dim=-1) - (p * lg).sum(-1)
                    ent.index_put_((dst_r[r0:r0 + 2048], dst_c[r0:r0 + 2048]), e)
    return out, ent


def grpo_compute_loss(ref_logps, new_logps, old_logps, completion_mask, advantages, beta=0.0, loss_type="grpo",
                      epsilon_low=0.2, epsilon_high=0.2, delta=None, max_completion_length=None, num_items_in_batch=None,
                      num_processes=1, importance_sampling_level="token"):
    """TRL GRPOTrainer._compute_loss on per-token log-probs [B, L]. Returns (loss, mean completion length, mean KL,
    coef_1, completion_mask) -- what the reference's compute_loss logs (rl_replacements.py:2941-3005)."""
    mask = completion_mask.to(new_logps.dtype)
    if advantages.dim() == 1:
        advantages = advantages.unsqueeze(1)
    old = new_logps.detach() if old_logps is None else old_logps
    log_ratio = new_logps - old
    if importance_sampling_level == "sequence":
        log_w = ((log_ratio * mask).sum(-1) / mask.sum(-1).clamp(min=1.0)).unsqueeze(-1)
    elif importance_sampling_level == "token":
        log_w = log_ratio
    else:
        raise ValueError(f"importance_sampling_level {importance_sampling_level!r}: 'token' or 'sequence'")
    coef_1 = torch.exp(log_w)
    coef_2 = torch.clamp(coef_1, 1 - epsilon_low, 1 + epsilon_high)
    c1 = torch.clamp(coef_1, max=delta) if delta is not None else coef_1
    per_token = -torch.min(c1 * advantages, coef_2 * advantages)
    mean_kl = torch.zeros((), device=new_logps.device)
    if beta != 0.0:
        if ref_logps is None:
            raise ValueError("beta != 0 needs the reference model's per-token log-probs")
        d = ref_logps - new_logps
        kl = torch.exp(d) - d - 1                                 # k3 estimator
        per_token = per_token + beta * kl
        mean_kl = (kl * mask).sum() / mask.sum().clamp(min=1.0)
    if loss_type == "grpo":
        loss = ((per_token * mask).sum(-1) / mask.sum(-1).clamp(min=1.0)).mean()
    elif loss_type == "bnpo":
        loss = (per_token * mask).sum() / mask.sum().clamp(min=1.0)
    elif loss_type == "dr_grpo":
        if max_completion_length is None:
            raise ValueError("dr_grpo normalises by max_completion_length")
        loss = (per_token * mask).sum() / (per_token.shape[0] * max_completion_length)
    elif loss_type == "dapo":
        norm = (num_items_in_batch / num_processes) if num_items_in_batch is not None else mask.sum().clamp(min=1.0)
        loss = (per_token * mask).sum() / norm
    else:
        raise ValueError(f"loss_type {loss_type!r}: grpo, bnpo, dr_grpo or dapo")
    return loss, mask.sum(-1).mean(), mean_kl, coef_1, completion_mask


def grpo_accumulated_loss(model, input_ids, attention_mask, logits_to_keep, completion_mask, advantages, old_logps=None,
                          ref_logps=None, beta=0.0, temperature=1.0, chunks=4, **loss_kwargs):
    """Policy log-probs (one packed forward, with gradient) + the GRPO objective."""
    new_logps, _ = get_per_token_logps_and_entropies(model, input_ids, attention_mask, logits_to_keep, temperature, chunks)
    return grpo_compute_loss(ref_logps, new_logps, old_logps, completion_mask, advantages, beta, **loss_kwargs)


def grpo_trainer__get_per_token_logps_and_entropies(self, model, input_ids, attention_mask, logits_to_keep, batch_size=None,
                                                    compute_entropy=False, compute_efficient=False, *args, **kwargs):
    """Method for TRL's GRPOTrainer (rl_replacements.py:1517-1700): (logps, entropies)."""
    if kwargs.get("pixel_values", None) is not None:
        raise NotImplementedError("GRPO over image inputs: compute the log-probs through FastVisionModel's forward")
    if compute_efficient:
        return None, None
    return get_per_token_logps_and_entropies(model, input_ids, attention_mask, logits_to_keep,
                                             temperature=getattr(self, "temperature", 1.0), compute_entropy=compute_entropy)


def grpo_trainer_compute_loss(self, model, inputs, return_outputs=False, num_items_in_batch=None):
    """Method for TRL's GRPOTrainer (rl_replacements.py:2622-3005), duck-typed on the attributes it reads: beta,
    epsilon_low / epsilon_high, temperature, importance_sampling_level, args.{loss_type, delta, max_completion_length},
    accelerator.num_processes, _metrics."""
    if return_outputs:
        raise ValueError("The GRPOTrainer does not support returning outputs")
    prompt_ids, prompt_mask = inputs["prompt_ids"], inputs["prompt_mask"]
    completion_ids, completion_mask = inputs["completion_ids"], inputs["completion_mask"]
    input_ids = torch.cat([prompt_ids, completion_ids], dim=1)
    attention_mask = torch.cat([prompt_mask, completion_mask], dim=1)
    logits_to_keep = completion_ids.size(1)
    loss_mask = completion_mask
    if inputs.get("tool_mask", None) is not None:
        if inputs["tool_mask"].shape != completion_mask.shape:
            raise ValueError("tool_mask/env_mask must have the same shape as completion_mask")
        loss_mask = completion_mask * inputs["tool_mask"].to(device=completion_mask.device, dtype=completion_mask.dtype)
    args = getattr(self, "args", None)
    acc = getattr(self, "accelerator", None)
    loss, length, mean_kl, coef_1, _ = grpo_accumulated_loss(
        model, input_ids, attention_mask, logits_to_keep, loss_mask, inputs["advantages"],
        old_logps=inputs.get("old_per_token_logps", None), ref_logps=inputs.get("ref_per_token_logps", None),
        beta=getattr(self, "beta", 0.0), temperature=getattr(self, "temperature", 1.0),
        loss_type=getattr(args, "loss_type", "grpo"), epsilon_low=getattr(self, "epsilon_low", 0.2),
        epsilon_high=getattr(self, "epsilon_high", 0.2), delta=getattr(args, "delta", None),
        max_completion_length=getattr(args, "max_completion_length", None),
        num_items_in_batch=inputs.get("num_items_in_batch", num_items_in_batch),
        num_processes=getattr(acc, "num_processes", 1),
        importance_sampling_level=getattr(self, "importance_sampling_level", "token"))
    metrics = getattr(self, "_metrics", None)
    if metrics is not None:
        mode = "eval" if getattr(getattr(self, "control", None), "should_evaluate", False) else "train"
        bucket = metrics[mode] if "train" in metrics else metrics
        bucket["completion_length"].append(float(length))
        bucket["kl"].append(float(mean_kl))
        adv = inputs["advantages"].unsqueeze(1) if inputs["advantages"].dim() == 1 else inputs["advantages"]
        m = loss_mask.to(coef_1.dtype)
        lo = ((coef_1 < 1 - getattr(self, "epsilon_low", 0.2)) & (adv < 0)).to(coef_1.dtype)
        hi = ((coef_1 > 1 + getattr(self, "epsilon_high", 0.2)) & (adv > 0)).to(coef_1.dtype)
        if coef_1.shape[1] == m.shape[1]:
            bucket["clip_ratio/region_mean"].append(float(((lo + hi) * m).sum() / m.sum().clamp(min=1.0)))
    return loss


def patch_grpo_trainer():
    """Installs the two methods on trl.GRPOTrainer when TRL is importable (the reference rewrites the trainer's source,
    rl.py / RL_FUNCTIONS["grpo_trainer"]); returns True when patched."""
    try:
        from trl import GRPOTrainer
    except Exception:
        return False
    GRPOTrainer._get_per_token_logps_and_entropies = grpo_trainer__get_per_token_logps_and_entropies
    GRPOTrainer.compute_loss = grpo_trainer_compute_loss
    return True


# ---- DPO ---------------------------------------------------------------------------------------------------------------
def dpo_sequence_logps(model, input_ids, attention_mask, completion_mask, temperature=1.0, chunks=4):
    """sum over the completion tokens of log p(token | prefix): [B]. `completion_mask` [B, L] marks the completion columns
    (the last L of input_ids); what TRL's DPOTrainer.concatenated_forward reduces its logits to."""
    L = completion_mask.shape[1]
    lp, _ = get_per_token_logps_and_entropies(model, input_ids, attention_mask, L, temperature, chunks)
    return (lp * completion_mask.to(lp.dtype)).sum(-1)


def dpo_loss(policy_chosen_logps, policy_rejected_logps, ref_chosen_logps, ref_rejected_logps, beta=0.1,
             label_smoothing=0.0, loss_type="sigmoid"):
    """DPO (Rafailov et al.) as TRL's DPOTrainer.dpo_loss computes it: (losses [B], chosen rewards, rejected rewards)."""
    logits = (policy_chosen_logps - policy_rejected_logps) - (ref_chosen_logps - ref_rejected_logps)
    if loss_type == "sigmoid":
        ls = torch.nn.functional.logsigmoid
        losses = -ls(beta * logits) * (1 - label_smoothing) - ls(-beta * logits) * label_smoothing
    elif loss_type == "hinge":
        losses = torch.relu(1 - beta * logits)
    elif loss_type == "ipo":
        losses = (logits - 1 / (2 * beta)) ** 2
    else:
        raise ValueError(f"loss_type {loss_type!r}: sigmoid, hinge or ipo")
    return (losses, beta * (policy_chosen_logps - ref_chosen_logps).detach(),
            beta * (policy_rejected_logps - ref_rejected_logps).detach())
