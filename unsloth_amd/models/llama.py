"""Llama-family fast path over HuggingFace transformers modules; mirror of the training half of
unsloth/models/llama.py.

  RopeTables                         LlamaRotaryEmbedding cache (llama.py:1775-1914), llama3 band scaling
                                     (:1688-1717), linear scaling (:1917-1945); rebuilt by us after load
                                     (never trust the loaded buffer), grows in 8192 steps
  LlamaAttention_fast_forward        llama.py:671-770
  LlamaDecoderLayer_fast_forward     llama.py:774-851 (training branch :823-844)
  LlamaModel_fast_forward            llama.py:866-1245 (embed -> layers (checkpointed) -> final norm)
  CausalLM_fast_forward              llama.py:1370-1590 (fused linear-CE branch :1466-1523, logits branch :1525-1562)
  FastLlamaModel                     pre_patch (:2288-2320), from_pretrained (:2323-3058),
                                     get_peft_model (:3061-3597), patch_peft_model (:3599-3821),
                                     for_training / for_inference (:3824-3929)

The composition reads HF module ATTRIBUTES (q_proj, input_layernorm.weight, ...) and never calls HF's
own layer forwards on the training path, so it is insensitive to transformers' internal attention /
cache / mask API (the installed 5.15 is above the reference's ceiling).
Attention is the hand-written causal GQA flash kernel pair of csrc/attention.hip (kernels/attention.py);
torch SDPA only serves key-padding masks with holes, small plain-causal batches of small-head models and G > 8.
"""
import math
import os
from types import MethodType
from typing import Optional

import torch
import torch.nn.functional as F
from transformers.modeling_outputs import CausalLMOutputWithPast

from ..kernels import (
    apply_lora_mlp_swiglu,
    apply_lora_o,
    apply_lora_qkv,
    fast_cross_entropy_loss,
    fast_rms_layernorm,
    fast_mrope_embedding,
    fast_rope_embedding,
    unsloth_fused_ce_loss,
)
from ..kernels.utils import invalidate_cast_cache, lora_linear_forward
from ..kernels.rms_layernorm import fast_add_rms_layernorm
from ..utils.packing import (
    get_packed_info_from_kwargs,
    mask_packed_boundary_labels,
    mask_packed_sequence_boundaries,
)
from .. import lora as _lora
from .. import nf4 as _nf4
from ..kernels import attention as _flash
from . import fast_layer as _fast_layer

_USE_FLASH = True
_FUSED_RESIDUAL = True
CHECK_POSITIONS = False        # True: assert on the host that packed position_ids restart where the documents do (a sync per step)

__version__ = "0.1.0"


# ------------------------------------------------------------------------------------------------
# Rotary tables (a5)
def _rope_params(config):
    rp = getattr(config, "rope_parameters", None) or {}
    rs = getattr(config, "rope_scaling", None) or {}
    theta = rp.get("rope_theta", None) or getattr(config, "rope_theta", None) or 10000.0
    kind = rp.get("rope_type", None) or rs.get("rope_type", None) or rs.get("type", None) or "default"
    merged = dict(rs)
    merged.update(rp)
    return float(theta), kind, merged


def _mrope_section(config):
    """(s_t, s_h, s_w) rotary pairs per position stream of a Qwen2-VL style config, else None."""
    rp = getattr(config, "rope_parameters", None) or {}
    rs = getattr(config, "rope_scaling", None) or {}
    sec = rp.get("mrope_section", None) or rs.get("mrope_section", None)
    return tuple(int(x) for x in sec) if sec else None


def compute_inv_freq(config):
    """(inv_freq fp32 [dim/2], attention_scaling, time_divisor) of the config's RoPE variant.

    default : theta^(-2i/dim) from an int64 arange cast to float (llama.py:1851-1855).
    llama3  : Llama-3.1's three frequency bands (llama.py:1688-1717; transformers ROPE_INIT_FUNCTIONS["llama3"]):
              wavelengths shorter than original_ctx/high_freq_factor keep their frequency, longer than
              original_ctx/low_freq_factor are slowed down by `factor`, and the band in between is interpolated.
              Written as ONE blend with a clamped weight -- w = 1 keeps, w = 0 divides -- which is bit-identical to
              the reference's select-of-three (0*x + 1*y == y and the band edges give w exactly 0 or 1).
    linear  : positions are DIVIDED by `factor` when the table is built (llama.py:1917-1945: `t / scaling_factor`,
              not inv_freq / factor as transformers does: the two round differently).
    Anything else (yarn / longrope / dynamic) is outside the hot-path scope and raises."""
    theta, kind, p = _rope_params(config)
    dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
    dim = int(dim * getattr(config, "partial_rotary_factor", 1.0))
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    attention_scaling, time_divisor = 1.0, 1.0
    if kind == "llama3":
        factor = p.get("factor", 8.0)
        low, high = p.get("low_freq_factor", 1.0), p.get("high_freq_factor", 4.0)
        ctx = p.get("original_max_position_embeddings", 8192)
        if low == high:
            raise ValueError("llama3 rope scaling needs low_freq_factor != high_freq_factor")
        periods_in_ctx = ctx / (2 * math.pi / inv_freq)            # how often the component wraps inside the old context
        w = ((periods_in_ctx - low) / (high - low)).clamp_(0.0, 1.0)
        inv_freq = (1 - w) * inv_freq / factor + w * inv_freq
    elif kind == "linear":
        time_divisor = float(p.get("factor", 1.0))
    elif kind not in ("default", None):
        raise NotImplementedError(f"rope_type {kind!r} (yarn/longrope/dynamic) is outside the hot path scope")
    return inv_freq, attention_scaling, time_divisor


class RopeTables:
    """cos/sin cache shared by all layers (llama.py:3002-3006), one per device."""

    def __init__(self, config):
        self.inv_freq, self.attention_scaling, self.time_divisor = compute_inv_freq(config)
        self.max_position_embeddings = getattr(config, "max_position_embeddings", 4096)
        self.current_rope_size = 0
        self._cache = {}

    def _build(self, seq_len, device, dtype):
        t = torch.arange(seq_len, dtype=torch.int64).float()
        if self.time_divisor != 1.0:
            t = t / self.time_divisor                           # linear scaling (llama.py:1941-1943)
        freqs = torch.outer(t, self.inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        cos = (emb.cos() * self.attention_scaling).to(dtype=dtype, device=device)
        sin = (emb.sin() * self.attention_scaling).to(dtype=dtype, device=device)
        return cos, sin

    def get(self, seq_len, device, dtype):
        key = (str(device), dtype)
        ent = self._cache.get(key)
        if ent is None or ent[0].shape[0] < seq_len:
            size = max(min(4 * 8192, self.max_position_embeddings), ((seq_len + 8191) // 8192) * 8192)
            ent = self._build(size, device, dtype)                  # grows in 8192 steps (:1906-1914)
            self._cache[key] = ent
            self.current_rope_size = size
        return ent


# ------------------------------------------------------------------------------------------------
def original_apply_qkv(self, X):
    return self.q_proj(X), self.k_proj(X), self.v_proj(X)


def original_apply_o(self, X):
    return self.o_proj(X)


_BAND_CACHE = {}     # device -> (seq_lengths tensor, (B, T, window), band): one band per batch, reused by all layers


def _attention_band(seq_info, B, T, window, device):
    """(lo, hi) int32 band of the packed / windowed causal mask, built once per batch like the reference's
    _SDPA_MASK_CACHE (utils/packing.py:657-693) -- but 8 bytes per token instead of a dense T x T mask."""
    seq_lengths = seq_info[0] if seq_info is not None else None
    ent = _BAND_CACHE.get(device)
    if ent is not None and ent[0] is seq_lengths and ent[1] == (B, T, window):
        return ent[2]
    band = _flash.attention_band(T, batch=B, seq_lengths=seq_lengths, sliding_window=window, device=device)
    _BAND_CACHE[device] = (seq_lengths, (B, T, window), band)
    return band


def _attention(Q, K, V, seq_info, attention_mask, sliding_window=None):
    """causal (GQA, packed, windowed) attention. Q [B,Hq,T,D], K/V [B,Hk,T,D] (strided views of the [B,T,H,D]
    projection outputs) -> [B, T, Hq*D].
    Causal batches -- plain, packed (block-diagonal over `seq_info` documents; right- / left-padded rows arrive
    here as documents too, see LlamaModel_fast_forward) or sliding-window -- take the hand-written CDNA4 kernels
    (kernels/attention.py), which read the [B,T,H,D] memory directly and write the o_proj input layout: no
    transposes, no copies, no dense mask. What is left for torch SDPA (run_attention's SDPA branch,
    attention_dispatch.py:560-617): key-padding masks with holes, more than 8 query heads per KV head, head dims above 128
    or not a multiple of 8."""
    B, Hq, T, D = Q.shape
    window = sliding_window if (sliding_window is not None and 0 < sliding_window < T) else None   # mistral.py:116-120
    if attention_mask is None and _USE_FLASH:
        q, k, v = Q.transpose(1, 2), K.transpose(1, 2), V.transpose(1, 2)          # [B,T,H,D] views
        need_band = seq_info is not None or window is not None
        # shapes the kernels take as they are: head_dim a multiple of 8 up to 128 (TinyLlama / Llama-3.2-1B: 64 -- no padded copies
        # since round 6: forward + backward 0.78 ms against SDPA's 1.35 ms at 4 x 2048 tokens, 0.35 vs 0.68 at 1 x 2048, 0.26 vs
        # 0.25 at 4 x 512, profiles/r06zf_attention_d64_vs_sdpa.jsonl), 1 .. 8 query heads per KV head (Qwen2.5-7B / Qwen2-VL-7B's
        # 28 on 4 included). Head dims that are not a multiple of 8 run on the same kernels zero-padded when the batch is packed /
        # windowed / padded or large enough to amortise the copies -- nothing builds a dense [T, T] mask.
        if _flash.native(q, k, v) or (_flash.supported(q, k, v) and (need_band or B * T >= 4096)):
            band = _attention_band(seq_info, B, T, window, Q.device) if need_band else None
            return _flash.flash_attention(q, k, v, None, band).reshape(B, T, Hq * D)
    # ---- the library kernel (ONE call site): plain causal without a mask tensor, everything else with a dense additive mask
    #      built from the same (lo, hi) band the flash kernels take -- one [T, T] block per batch row
    mask = None
    if not (seq_info is None and attention_mask is None and window is None):
        pos = torch.arange(T, device=Q.device)
        allowed = (pos[:, None] >= pos[None, :])[None]                          # causal, [1, T(q), T(key)]
        if seq_info is not None or window is not None:
            lo, _ = _attention_band(seq_info, B, T, window, Q.device)
            allowed = allowed & (pos[None, None, :] >= lo[:, :, None])
        allowed = allowed[:, None]
        if attention_mask is not None:
            allowed = allowed & attention_mask.to(torch.bool)[:, None, None, :]
        mask = torch.zeros(allowed.shape, dtype=Q.dtype, device=Q.device).masked_fill_(~allowed, float("-inf"))
    A = F.scaled_dot_product_attention(Q, K, V, attn_mask=mask, is_causal=mask is None, enable_gqa=True)
    return A.transpose(1, 2).reshape(B, T, Hq * D)


def LlamaAttention_fast_forward(self, hidden_states, cos, sin, rope_position_ids=None, seq_info=None,
                                attention_mask=None):
    """llama.py:671-770 (training path: no KV cache)."""
    bsz, q_len, _ = hidden_states.size()
    cfg = self.config
    n_heads, n_kv_heads = cfg.num_attention_heads, cfg.num_key_value_heads
    head_dim = self.head_dim
    Q, K, V = self.apply_qkv(self, hidden_states)
    Q = Q.view(bsz, q_len, n_heads, head_dim).transpose(1, 2)
    K = K.view(bsz, q_len, n_kv_heads, head_dim).transpose(1, 2)
    V = V.view(bsz, q_len, n_kv_heads, head_dim).transpose(1, 2)
    if isinstance(rope_position_ids, tuple):                             # multimodal RoPE: (positions3, mrope_section)
        Q, K = fast_mrope_embedding(Q, K, cos, sin, rope_position_ids[0], rope_position_ids[1])
    else:
        Q, K = fast_rope_embedding(Q, K, cos, sin, rope_position_ids)    # in place on the strided views
    attn_output = _attention(Q, K, V, seq_info, attention_mask, getattr(cfg, "sliding_window", None))
    return self.apply_o(self, attn_output)


def LlamaDecoderLayer_fast_forward(self, hidden_states, cos, sin, rope_position_ids=None, seq_info=None,
                                   attention_mask=None):
    """llama.py:823-844."""
    residual = hidden_states
    hidden_states = fast_rms_layernorm(self.input_layernorm, hidden_states)
    hidden_states = LlamaAttention_fast_forward(self.self_attn, hidden_states, cos, sin, rope_position_ids,
                                                seq_info, attention_mask)
    if _FUSED_RESIDUAL:
        residual, hidden_states = fast_add_rms_layernorm(self.post_attention_layernorm, hidden_states, residual)
    else:
        hidden_states = residual + hidden_states
        residual = hidden_states
        hidden_states = fast_rms_layernorm(self.post_attention_layernorm, hidden_states)
    hidden_states = self.mlp(hidden_states)
    hidden_states = residual + hidden_states
    return hidden_states


def LlamaDecoderLayer_fused_residual_forward(self, residual, delta, cos, sin, rope_position_ids=None, seq_info=None,
                                             attention_mask=None):
    """The same layer with the residual stream carried as (residual, delta): hidden = residual + delta is never
    formed by a separate pass -- each add is fused into the norm that follows it (kernels/rms_layernorm.py
    Fast_Add_RMS_Layernorm), also across the layer boundary. Returns (residual', delta') for the next layer."""
    if delta is None:
        hidden = residual
        x = fast_rms_layernorm(self.input_layernorm, hidden)
    else:
        hidden, x = fast_add_rms_layernorm(self.input_layernorm, delta, residual)
    attn = LlamaAttention_fast_forward(self.self_attn, x, cos, sin, rope_position_ids, seq_info, attention_mask)
    hidden, x = fast_add_rms_layernorm(self.post_attention_layernorm, attn, hidden)
    return hidden, self.mlp(x)


def LlamaModel_fast_forward(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None,
                            **kwargs):
    """llama.py:866-1245, training path. `self` is the HF LlamaModel."""
    invalidate_cast_cache()          # cached bf16 copies of the LoRA factors live for ONE forward/backward
    if inputs_embeds is None:
        inputs_embeds = self.embed_tokens(input_ids)
    dtype = _model_dtype(self)
    hidden_states = inputs_embeds.to(dtype)                                        # :955-958
    bsz, q_len, _ = hidden_states.shape
    seq_info = get_packed_info_from_kwargs(kwargs, hidden_states.device)
    # padding masks that are all ones are dropped in training (:1017-1019)
    if attention_mask is not None and (attention_mask.dim() != 2 or bool(torch.all(attention_mask != 0))):
        attention_mask = None
    if attention_mask is not None and seq_info is None and attention_mask.shape == (bsz, q_len):
        # right- / left-padded rows are packed documents [padding | tokens | padding]: the band kernels take them,
        # nothing builds a dense mask (rows with holes keep the SDPA branch)
        docs = _flash.padding_mask_documents(attention_mask.to(hidden_states.device))
        if docs is not None:
            seq_info, attention_mask = (docs, None, None), None
    rope_position_ids = None
    mrope = None
    if position_ids is not None and position_ids.dim() == 3 and position_ids.shape[0] == 3:
        # Qwen2-VL style multimodal positions [3, B, T] (temporal, height, width): the text tower of BASELINE config 4
        section = _mrope_section(self.config)
        if section is None:
            raise ValueError("position_ids of shape [3, B, T] need `mrope_section` in the config's rope parameters")
        mrope = (position_ids.to(device=hidden_states.device, dtype=torch.int32).contiguous(), section)
        position_ids = None
    if position_ids is not None:
        rope_position_ids = position_ids.to(device=hidden_states.device, dtype=torch.int32)    # :945-948
        if rope_position_ids.dim() == 1:
            rope_position_ids = rope_position_ids.unsqueeze(0)
        if rope_position_ids.shape[0] != bsz:
            rope_position_ids = rope_position_ids.expand(bsz, -1)
        rope_position_ids = rope_position_ids.reshape(-1)
    tables = _rope_tables(self)
    cos, sin = tables.get(max(q_len, 1), hidden_states.device, dtype)
    if rope_position_ids is not None and CHECK_POSITIONS:
        assert int(rope_position_ids.max()) < cos.shape[0]
    gc = bool(getattr(self, "gradient_checkpointing", False)) and self.training and torch.is_grad_enabled()
    policy = getattr(self, "_unsloth_amd_layer_policy", None)
    if mrope is not None:
        rope_position_ids = mrope                        # (positions3, section): every composition below takes the pair
    if policy is not None and self.training and torch.is_grad_enabled() \
            and all(_fast_layer.layer_supported(l, hidden_states, attention_mask) for l in self.layers):
        # use_gradient_checkpointing="unsloth": every layer is ONE manual-autograd Function that keeps what the
        # policy names and recomputes the rest in its backward (models/fast_layer.py)
        sw = getattr(self.config, "sliding_window", None)
        window = sw if (sw is not None and 0 < sw < q_len) else None
        band = None if (seq_info is None and window is None) else \
            _attention_band(seq_info, bsz, q_len, window, hidden_states.device)
        residual, delta = hidden_states, None
        step_decode = False
        if policy == _fast_layer.AUTO:
            policy = _fast_layer.auto_policy(self, hidden_states)
            step_decode = bool(getattr(self, "_uamd_step_decode", False))
        for li, layer in enumerate(self.layers):
            residual, delta = _fast_layer.decoder_layer_forward(layer, residual, delta, cos, sin, rope_position_ids,
                                                                band, _fast_layer.policy_for_layer(policy, li), step_decode)
        return fast_add_rms_layernorm(self.norm, delta, residual)[1]
    if gc and not hidden_states.requires_grad:
        hidden_states.requires_grad_(True)      # reentrant checkpoint needs an input that requires grad
    if not gc and _FUSED_RESIDUAL:
        # residual stream carried as (residual, delta): every `residual + x` is fused into the norm after it
        residual, delta = hidden_states, None
        for layer in self.layers:
            residual, delta = LlamaDecoderLayer_fused_residual_forward(layer, residual, delta, cos, sin,
                                                                       rope_position_ids, seq_info, attention_mask)
        if delta is None:
            return fast_rms_layernorm(self.norm, residual)
        return fast_add_rms_layernorm(self.norm, delta, residual)[1]                # :1228
    for layer in self.layers:
        if gc:
            # llama.py:1169-1193: reentrant, no RNG state (dropout is 0 on this path)
            hidden_states = torch.utils.checkpoint.checkpoint(
                _layer_fn(layer, cos, sin, rope_position_ids, seq_info, attention_mask), hidden_states,
                use_reentrant=True, preserve_rng_state=False)
        else:
            hidden_states = LlamaDecoderLayer_fast_forward(layer, hidden_states, cos, sin, rope_position_ids,
                                                           seq_info, attention_mask)
    return fast_rms_layernorm(self.norm, hidden_states)                            # :1228


def _layer_fn(layer, cos, sin, pos, seq_info, mask):
    def custom_forward(h):
        return LlamaDecoderLayer_fast_forward(layer, h, cos, sin, pos, seq_info, mask)
    return custom_forward


def _model_dtype(model):
    dt = getattr(model, "_unsloth_amd_dtype", None)
    if dt is None:
        dt = getattr(model.config, "dtype", None) or getattr(model.config, "torch_dtype", None) or torch.bfloat16
        if isinstance(dt, str):
            dt = getattr(torch, dt)
    return dt


def _rope_tables(model):
    t = getattr(model, "_unsloth_amd_rope", None)
    if t is None:
        t = RopeTables(model.config)
        model._unsloth_amd_rope = t
    return t


_LOGITS_MESSAGE = ("Unsloth: logits are not materialised on the fused cross-entropy path. "
                   "Set UNSLOTH_RETURN_LOGITS=1 to get them.")


def _no_logits(*args, **kwargs):
    raise NotImplementedError(_LOGITS_MESSAGE)


class _EmptyLogits:
    """Stand-in for `logits` on the fused-CE path (models/_utils.py:3612-3652). USING it raises; merely LOOKING at it must
    not: accelerate's bf16 wrapper probes `hasattr(x, "dtype") and x.dtype in (fp16, bf16)` on every field of the model
    output (accelerate/utils/operations.py convert_to_fp32), and a distributed Trainer pickles / compares it. So an
    attribute is a callable that raises when called, `.to(...)` gives None, indexing raises, and the object pickles to
    the singleton."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)                 # protocol probes (copy, pickle, numpy) get the truthful answer
        return (lambda *a, **k: None) if name == "to" else _no_logits

    def __getitem__(self, item):
        raise NotImplementedError(_LOGITS_MESSAGE)

    def __iter__(self):
        raise NotImplementedError(_LOGITS_MESSAGE)

    def __reduce__(self):
        return (_empty_logits, ())

    def __eq__(self, other):
        return type(other).__name__ == "_EmptyLogits"

    __hash__ = object.__hash__

    def __repr__(self):
        return "EMPTY_LOGITS"


def _empty_logits():
    return EMPTY_LOGITS


EMPTY_LOGITS = _EmptyLogits()


def CausalLM_fast_forward(original_forward):
    """llama.py:1370-1590. Returns the function installed as LlamaForCausalLM.forward."""

    def _CausalLM_fast_forward(self, input_ids=None, attention_mask=None, position_ids=None,
                               past_key_values=None, inputs_embeds=None, labels=None, use_cache=None,
                               logits_to_keep=0, num_logits_to_keep=0, return_dict=None, **kwargs):
        fast = getattr(self, "_unsloth_amd_fast", False)
        if (not fast) or past_key_values is not None or (use_cache and not self.training and labels is None):
            # decode / generation: out of scope, HF's own forward over our Linear4bit/LoraLayer modules
            return original_forward(self, input_ids=input_ids, attention_mask=attention_mask,
                                    position_ids=position_ids, past_key_values=past_key_values,
                                    inputs_embeds=inputs_embeds, labels=labels, use_cache=use_cache,
                                    logits_to_keep=logits_to_keep, **kwargs)
        hidden_states = LlamaModel_fast_forward(self.model, input_ids=input_ids, attention_mask=attention_mask,
                                                position_ids=position_ids, inputs_embeds=inputs_embeds, **kwargs)
        lm_head = self.lm_head.weight
        logit_softcapping = getattr(self.config, "final_logit_softcapping", 0) or 0
        logit_scaling = getattr(self.config, "logit_scale", 0) or 0
        if os.environ.get("UNSLOTH_RETURN_HIDDEN_STATES", "0") == "1":             # :1448-1457
            return CausalLMOutputWithPast(loss=None, logits=hidden_states)
        RETURN_LOGITS = os.environ.get("UNSLOTH_RETURN_LOGITS", "0") == "1"
        n_items = kwargs.get("num_items_in_batch", None)
        if n_items is None:
            n_items = kwargs.get("n_items", None)
        can_fuse = (labels is not None and not RETURN_LOGITS and self.lm_head.bias is None and not logit_scaling
                    and (not lm_head.requires_grad or lm_head.dtype == hidden_states.dtype))
        if can_fuse:
            labels = mask_packed_boundary_labels(labels.to(hidden_states.device), kwargs.get("packed_seq_lengths"))
            loss = unsloth_fused_ce_loss(trainer=None, hidden_states=hidden_states, lm_head_weight=lm_head,
                                         lm_head_bias=None, labels=labels, mask=None, n_items=n_items,
                                         scaling=getattr(self, "accelerator_scaler", None), target_gb=None,
                                         torch_compile=True, logit_softcapping=logit_softcapping)
            if return_dict is False:
                return (loss, EMPTY_LOGITS)
            return CausalLMOutputWithPast(loss=loss, logits=EMPTY_LOGITS)
        keep = max(int(num_logits_to_keep or 0), int(logits_to_keep or 0) if isinstance(logits_to_keep, int) else 0)
        hs = hidden_states[:, -keep:, :] if keep else hidden_states
        if hs.dtype in (torch.bfloat16, torch.float16) and lm_head.dtype == hs.dtype:
            # through autograd Functions (the MFMA GEMM forward, NN-form dX, dW for a trainable head): a bare kernel call
            # here would cut the graph -- a loss computed from these logits must still train the model
            from ..kernels.fast_dense import Dense_W
            from ..kernels.fast_lora import LoRA_W
            if lm_head.requires_grad or (self.lm_head.bias is not None and self.lm_head.bias.requires_grad):
                logits = Dense_W.apply(hs, lm_head, self.lm_head.bias)
            elif self.lm_head.bias is not None:
                logits = LoRA_W.apply(hs, lm_head, None, None, None, None, self.lm_head.bias)
            else:
                logits = LoRA_W.apply(hs, lm_head, None, None, None, None)
        else:
            logits = self.lm_head(hs)
        loss = None
        if labels is not None:
            labels = labels.to(logits.device)
            shift_labels = torch.empty_like(labels)                                # :1545-1551
            shift_labels[..., :-1] = labels[..., 1:]
            shift_labels[..., -1] = -100
            mask_packed_sequence_boundaries(shift_labels, kwargs.get("packed_seq_lengths"))
            loss = fast_cross_entropy_loss(logits=logits, labels=shift_labels, logit_softcapping=logit_softcapping,
                                           logit_scaling=logit_scaling, n_items=n_items)
        else:
            if logit_scaling:
                logits = logit_scaling * logits
            if logit_softcapping:
                logits = logit_softcapping * torch.tanh(logits / logit_softcapping)
        if return_dict is False:
            return (loss, logits) if loss is not None else (logits,)
        return CausalLMOutputWithPast(loss=loss, logits=logits)

    return _CausalLM_fast_forward


# ------------------------------------------------------------------------------------------------
# Class-level patch points (llama.py:2300-2319). The reference assigns its fast forwards to the HF classes so that
# code which calls a LAYER directly (not the CausalLM) still runs the fused kernels. transformers 5.x passes
# different arguments to these methods than the reference's pinned versions, so each patch below is an adapter with
# HF's current signature around the same fast composition; instances that were not prepared by FastLlamaModel
# (no `apply_qkv` hook, no rope tables) and cached / decoding calls fall through to the original method.
_PATCHED = {}


def _patch_method(cls, name, make):
    key = (cls, name)
    if key not in _PATCHED:
        original = getattr(cls, name)
        _PATCHED[key] = original
        setattr(cls, name, make(original))


def unpatch_all():
    """Restore every class-level method replaced by pre_patch()."""
    for (cls, name), original in list(_PATCHED.items()):
        setattr(cls, name, original)
    _PATCHED.clear()


def _hf_attention_forward(original):
    def forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
        if (not hasattr(self, "apply_qkv") or past_key_values is not None or position_embeddings is None
                or (attention_mask is not None and attention_mask.dim() != 2)
                or hidden_states.dtype not in (torch.bfloat16, torch.float16) or not hidden_states.is_cuda):
            return original(self, hidden_states, position_embeddings=position_embeddings,
                            attention_mask=attention_mask, past_key_values=past_key_values, **kwargs)
        cos, sin = position_embeddings                   # HF hands over the rows already gathered per token [B,T,D]
        B, T, _ = hidden_states.shape
        D = cos.shape[-1]
        cos = cos.expand(B, T, D).reshape(B * T, D)
        sin = sin.expand(B, T, D).reshape(B * T, D)
        idx = torch.arange(B * T, dtype=torch.int32, device=hidden_states.device)   # table row == token
        if attention_mask is not None and bool(torch.all(attention_mask != 0)):
            attention_mask = None
        out = LlamaAttention_fast_forward(self, hidden_states, cos, sin, idx, None, attention_mask)
        return out, None
    return forward


def _hf_decoder_layer_forward(original):
    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                position_embeddings=None, **kwargs):
        if not hasattr(self.self_attn, "apply_qkv") or past_key_values is not None or position_embeddings is None:
            return original(self, hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                            past_key_values=past_key_values, use_cache=use_cache,
                            position_embeddings=position_embeddings, **kwargs)
        residual = hidden_states
        x = fast_rms_layernorm(self.input_layernorm, hidden_states)
        x, _ = self.self_attn(hidden_states=x, attention_mask=attention_mask, position_ids=position_ids,
                              past_key_values=None, use_cache=use_cache, position_embeddings=position_embeddings,
                              **kwargs)
        residual, x = fast_add_rms_layernorm(self.post_attention_layernorm, x, residual)
        return residual + self.mlp(x)
    return forward


def _hf_model_forward(original):
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, use_cache=None, **kwargs):
        fast = (getattr(self, "_unsloth_amd_rope", None) is not None and past_key_values is None
                and (self.training or not use_cache))
        if not fast:
            return original(self, input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                            past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache,
                            **kwargs)
        from transformers.modeling_outputs import BaseModelOutputWithPast
        h = LlamaModel_fast_forward(self, input_ids=input_ids, attention_mask=attention_mask,
                                    position_ids=position_ids, inputs_embeds=inputs_embeds, **kwargs)
        return BaseModelOutputWithPast(last_hidden_state=h, past_key_values=None)
    return forward


def PeftModel_fast_forward(self, input_ids=None, causal_mask=None, attention_mask=None, inputs_embeds=None,
                           labels=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                           task_ids=None, num_logits_to_keep=0, logits_to_keep=0, **kwargs):
    """llama.py:1594-1634: PEFT's wrapper forwards straight to the (patched) base model, dropping the adapter
    bookkeeping arguments the fused path has no use for."""
    return self.base_model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                           labels=labels, return_dict=return_dict, num_logits_to_keep=num_logits_to_keep,
                           logits_to_keep=logits_to_keep, **kwargs)


def _patch_causal_lm_class(cls):
    _patch_method(cls, "forward", CausalLM_fast_forward)


def _patch_architecture(mod, prefix):
    """Attention / DecoderLayer / Model / ForCausalLM of one modeling module (llama, mistral, qwen2 share the code)."""
    for suffix, make in (("Attention", _hf_attention_forward), ("DecoderLayer", _hf_decoder_layer_forward),
                         ("Model", _hf_model_forward)):
        cls = getattr(mod, prefix + suffix, None)
        if cls is not None:
            _patch_method(cls, "forward", make)
    cls = getattr(mod, prefix + "ForCausalLM", None)
    if cls is not None:
        _patch_causal_lm_class(cls)


class FastLlamaModel:
    """Patcher for Llama-architecture models (Llama, Mistral, Qwen2 share it: qwen2.py:38-97)."""

    @staticmethod
    def pre_patch():
        """llama.py:2288-2320: class-level forward replacement + HF RMSNorm class swap + loss mapping."""
        import importlib
        from ..kernels import patch_loss_functions, patch_rms_layernorm
        for modname, prefix in (("llama", "Llama"), ("mistral", "Mistral"), ("qwen2", "Qwen2")):
            try:
                mod = importlib.import_module(f"transformers.models.{modname}.modeling_{modname}")
            except Exception:
                continue
            _patch_architecture(mod, prefix)
        try:                                             # real PEFT, when installed (not in this image)
            import peft
            _patch_method(peft.PeftModelForCausalLM, "forward", lambda original: PeftModel_fast_forward)
        except Exception:
            pass
        _patch_method(_lora.PeftModelForCausalLM, "forward", lambda original: PeftModel_fast_forward)
        patch_rms_layernorm()
        patch_loss_functions()

    @staticmethod
    def post_load(model, max_seq_length, dtype):
        """Per-instance setup after weights exist: default hooks (llama.py:2851-2853), shared rope
        tables rebuilt from config (:3002-3006), bookkeeping."""
        inner = model.model
        inner._unsloth_amd_dtype = dtype
        inner._unsloth_amd_rope = RopeTables(model.config)
        for m in model.modules():
            # the decode writes quant_state.dtype and the GEMM reads it as the activation dtype: keep them equal,
            # whatever the checkpoint was stamped with (the reference does the same after load, granite.py:586-596)
            if isinstance(m, _nf4.Linear4bit) and dtype is not None:
                m.weight.quant_state.dtype = dtype
                m.compute_dtype = dtype
        for layer in inner.layers:
            layer.self_attn.apply_qkv = original_apply_qkv
            layer.self_attn.apply_o = original_apply_o
        model._unsloth_amd_fast = True
        model.max_seq_length = max_seq_length
        m = model
        while hasattr(m, "model"):
            m.max_seq_length = max_seq_length
            m = m.model
        m.max_seq_length = max_seq_length
        # llama.py:2988-3000 + loader.py:1114: the stock-Trainer glue -- num_items_in_batch reaches the model and is counted
        # over the shifted labels, no nn.DataParallel around a replica, rotary inv_freq buffers out of DDP's broadcasts
        from ._utils import prepare_for_trainer
        prepare_for_trainer(model)
        return model

    @staticmethod
    def get_peft_model(model, r=16, target_modules=("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj",
                                                    "up_proj", "down_proj"),
                       lora_alpha=16, lora_dropout=0.0, bias="none", layers_to_transform=None,
                       layers_pattern=None, use_gradient_checkpointing="unsloth", random_state=3407,
                       max_seq_length=2048, use_rslora=False, modules_to_save=None, init_lora_weights=True,
                       loftq_config={}, temporary_location="_unsloth_temporary_saved_buffers", qat_scheme=None,
                       **kwargs):
        """llama.py:3061-3597."""
        if isinstance(model, _lora.PeftModelForCausalLM):
            # idempotent re-call with equal settings (:3145-3233), else TypeError (:3231)
            cfg = model.peft_config[model.active_adapter]
            same = (cfg.r == r and cfg.lora_alpha == lora_alpha and cfg.lora_dropout == lora_dropout
                    and sorted(cfg.target_modules) == sorted(target_modules))
            if same:
                from ._utils import exclude_rope_inv_freq_from_ddp
                return exclude_rope_inv_freq_from_ddp(model)                                # :3228
            raise TypeError("Unsloth: Your model already has LoRA adapters. Your new parameters are different.")
        if not isinstance(r, int) or r <= 0:
            raise TypeError(f"Unsloth: Rank of {str(r)} must be an integer larger than 0.")   # :3140-3143
        if bias != "none":
            raise NotImplementedError("bias != 'none' takes the slow PEFT path in the reference; not implemented")
        extra = [t for t in target_modules if t in ("lm_head", "embed_tokens")]
        if extra:
            # the reference moves these into PEFT's modules_to_save (trainable fp32 copies, llama.py:3290-3330); this PEFT
            # stand-in has no such wrapper -- refuse by name instead of wrapping them as LoRA layers
            raise NotImplementedError(f"target_modules {extra}: training lm_head / embed_tokens next to LoRA needs PEFT's "
                                      "modules_to_save (not in this build); full_finetuning=True trains them")
        torch.manual_seed(random_state)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(random_state)
        config = _lora.LoraConfig(r=r, lora_alpha=lora_alpha, target_modules=list(target_modules),
                                  lora_dropout=lora_dropout, bias=bias, use_rslora=use_rslora,
                                  modules_to_save=modules_to_save, init_lora_weights=init_lora_weights,
                                  layers_to_transform=layers_to_transform, loftq_config=loftq_config)
        peft_model = _lora.get_peft_model(model, config)
        return FastLlamaModel.patch_peft_model(peft_model, use_gradient_checkpointing)

    @staticmethod
    def patch_peft_model(model, use_gradient_checkpointing="unsloth"):
        """llama.py:3599-3821: enable GC, then per layer install the fused hooks iff dropout == 0,
        bias == 'none', bias-free base layers, no DoRA (:3695-3772); otherwise keep the defaults."""
        base = model.get_base_model() if hasattr(model, "get_base_model") else model
        inner = base.model
        FastLlamaModel.for_training(model, use_gradient_checkpointing)
        n_mlp = n_qkv = n_o = 0

        def ok(proj, bias_ok=False):
            # the reference's preconditions (:3695-3772) -- except that a FROZEN bias on an attention projection
            # (Qwen2's q/k/v) stays on the fused path here: it rides in the GEMM epilogue (uamd_gemm_group.bias)
            if not isinstance(proj, _lora.LoraLayer):
                return False
            ad = proj.active_adapters[0]
            bias = proj.base_layer.bias
            return (isinstance(proj.lora_dropout[ad], torch.nn.Identity)
                    and (bias is None or (bias_ok and not bias.requires_grad))
                    and not proj.use_dora[ad] and len(proj.lora_magnitude_vector) == 0)

        for layer in inner.layers:
            mlp, attn = layer.mlp, layer.self_attn
            if all(hasattr(mlp, n) and ok(getattr(mlp, n)) for n in ("gate_proj", "up_proj", "down_proj")):
                mlp.forward = MethodType(apply_lora_mlp_swiglu, mlp)                # :3725
                n_mlp += 1
            if all(ok(getattr(attn, n), True) for n in ("q_proj", "k_proj", "v_proj")):
                attn.apply_qkv = apply_lora_qkv                                     # :3748
                n_qkv += 1
            if ok(attn.o_proj, True):
                attn.apply_o = apply_lora_o                                         # :3766
                n_o += 1
        base._unsloth_amd_patched = (n_qkv, n_o, n_mlp)
        if os.environ.get("UNSLOTH_ENABLE_LOGGING", "0") == "1":
            print(f"Unsloth (MI355X) patched {len(inner.layers)} layers with {n_qkv} QKV layers, "
                  f"{n_o} O layers and {n_mlp} MLP layers.")                        # :3774-3777
        model.for_training = MethodType(FastLlamaModel.for_training, model)          # :3811-3817
        model.for_inference = MethodType(FastLlamaModel.for_inference, model)
        # :3575 + :3595 -- again, on the wrapper: the marker must sit on the object Trainer sees, and the buffers' fully
        # qualified names changed under the PEFT wrapper ("base_model.model. ...")
        from ._utils import prepare_for_trainer
        return prepare_for_trainer(model)

    @staticmethod
    def patch_full_finetune(model):
        """`full_finetuning=True` (loader.py:487-523; the reference leaves such a model to torch.nn.Linear + autograd):
        per layer install the dense manual-autograd blocks of kernels/fast_dense.py -- fused Q|K|V, o_proj, SwiGLU MLP,
        each with its weight gradient (uamd_gemm_tn_256) -- when the projections are plain 16-bit Linear modules."""
        from ..kernels.fast_dense import apply_dense_mlp_swiglu, apply_dense_o, apply_dense_qkv
        inner = model.model
        n_mlp = n_qkv = n_o = 0

        def ok(m):
            return (type(m) is torch.nn.Linear and m.weight.dtype in (torch.bfloat16, torch.float16)
                    and m.weight.is_cuda and m.in_features % 8 == 0 and m.out_features % 8 == 0)

        swiglu = getattr(model.config, "hidden_act", "silu") == "silu"
        for layer in inner.layers:
            mlp, attn = layer.mlp, layer.self_attn
            if swiglu and all(hasattr(mlp, n) and ok(getattr(mlp, n)) and getattr(mlp, n).bias is None
                              for n in ("gate_proj", "up_proj", "down_proj")):
                mlp.forward = MethodType(apply_dense_mlp_swiglu, mlp)
                n_mlp += 1
            if all(ok(getattr(attn, n)) for n in ("q_proj", "k_proj", "v_proj")):
                attn.apply_qkv = apply_dense_qkv
                n_qkv += 1
            if ok(attn.o_proj):
                attn.apply_o = apply_dense_o
                n_o += 1
        model._unsloth_amd_patched = (n_qkv, n_o, n_mlp)
        model.for_training = MethodType(FastLlamaModel.for_training, model)
        model.for_inference = MethodType(FastLlamaModel.for_inference, model)
        return model

    @staticmethod
    def for_training(model, use_gradient_checkpointing=True):
        """llama.py:3824-3885 + the mode selection of models/_utils.py:360-386.
          False      : every activation stays in HBM (288 GB), no recompute.
          True       : torch's reentrant per-layer checkpoint, what the reference gives for `True` (llama.py:1169-1193).
          "unsloth"  : the reference's smart mode is a fit-to-memory decision (models/_utils.py:360-386: offload layer
                       inputs to host RAM and re-run whole layers when VRAM is short). Here the same spelling resolves,
                       per call, to the LEAST-RECOMPUTE schedule that fits the free HBM (fast_layer.auto_schedule from the
                       batch size and the free memory: keep everything in as many layers as fit -- all 32 on an idle
                       288 GB part, i.e. the speed of `False` -- and fall back layer by layer to "attn" as memory gets
                       short). Fixed policies stay reachable by name: "unsloth:attn" (per layer keep the layer input,
                       Q/K/V + attention output and the post-attention residual, re-run only norm2 + the gate/up GEMM),
                       "unsloth:min" (layer input only: the memory of `True`), "unsloth:all" (everything), schedules
                       such as "unsloth:all*4,attn"; "unsloth:auto" is the explicit name of the default.
                       UNSLOTH_AMD_GC_POLICY overrides what the bare spelling means."""
        base = model.get_base_model() if hasattr(model, "get_base_model") else model
        policy = None
        mode = use_gradient_checkpointing
        if isinstance(mode, str) and mode.split(":")[0] == "unsloth":
            name = mode.split(":", 1)[1] if ":" in mode else os.environ.get("UNSLOTH_AMD_GC_POLICY", "auto")
            policy = _fast_layer.resolve_policy_spec(name)
        gc = bool(mode)
        base.model.gradient_checkpointing = gc
        base.model._unsloth_amd_layer_policy = policy
        from .. import nf4 as _nf4
        if policy != _fast_layer.AUTO and getattr(base.model, "_uamd_mirrors_auto", False):
            # decoded weight mirrors belong to the fit-to-memory spelling (nf4.py); a fixed mode asked for something else
            _nf4.set_resident(False, model=base.model)
            base.model._uamd_auto_policy = None
        for m in base.modules():
            if hasattr(m, "gradient_checkpointing"):
                m.gradient_checkpointing = gc
        base._unsloth_amd_fast = True
        if hasattr(model, "_old_generate"):                    # undo for_inference (llama.py:3873-3885)
            model.generate = model._old_generate
            del model._old_generate
        if hasattr(model, "_uamd_decode_engine"):
            del model._uamd_decode_engine
        model.train()
        return model

    @staticmethod
    def for_inference(model):
        """llama.py:3888-3929: eval mode, no checkpointing, and `model.generate` = the KV-cache decode engine
        (models/decode.py: unsloth_fast_generate, the counterpart of llama.py:2167-2259) for the architectures it covers
        (head_dim 128, SwiGLU); the HF `generate` stays reachable as `model._old_generate`, as in the reference."""
        from types import MethodType
        from . import decode as _decode
        base = model.get_base_model() if hasattr(model, "get_base_model") else model
        for m in base.modules():
            if hasattr(m, "gradient_checkpointing"):
                m.gradient_checkpointing = False
        model.eval()
        cfg = base.config
        head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        if head_dim == 128 and getattr(cfg, "hidden_act", "silu") == "silu" and not hasattr(model, "_old_generate"):
            if hasattr(model, "generate"):
                model._old_generate = model.generate
            model.generate = MethodType(_decode.unsloth_fast_generate, model)
        return model


def quantize_model_nf4_(model, blocksize=64, compress_statistics=True,
                        names=("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")):
    """Replace the decoder projections by frozen NF4 layers in place (what loading with
    BitsAndBytesConfig(nf4, double_quant) gives the reference, llama.py:2615-2626). lm_head and the
    embeddings stay 16-bit, as bitsandbytes' default skip list does."""
    for layer in model.model.layers:
        for parent in (layer.self_attn, layer.mlp):
            for n in names:
                lin = getattr(parent, n, None)
                if isinstance(lin, torch.nn.Linear):
                    setattr(parent, n, _nf4.Linear4bit.from_linear(lin, blocksize, compress_statistics))
    return model
