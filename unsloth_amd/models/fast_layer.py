"""One decoder layer as ONE manual-autograd Function with selective recompute: the MI355X answer to the reference's
`use_gradient_checkpointing="unsloth"` (hook site unsloth/models/llama.py:1169-1193, mode selection
models/_utils.py:360-386).

The reference's "unsloth" mode wraps every decoder layer in unsloth_zoo's offloaded checkpoint: keep only the layer
input (copied to host RAM), re-run the WHOLE layer in the backward. On a 288 GB part neither the PCIe round trip
nor most of that recompute is needed. This Function runs the layer's forward without building an autograd graph,
keeps per layer only what a policy names, and in the backward recomputes exactly the missing tensors before
walking the same manual backward chain the per-block Functions use (kernels/fast_lora.py):

  always kept   h0 = residual + delta (the layer input, 2 B/token/hidden -- what any checkpoint keeps), the two rms
                statistics, the rank-r products X A^T (fp32 [T, r], ~3.6 MB per layer at 8192 tokens), the
                attention log-sum-exp when Q/K/V are kept
  "qkv"         post-RoPE Q, K, V and the attention output          else: norm1 -> q/k/v GEMM -> RoPE -> flash fwd again
  "h1"          the residual stream after attention                  else: o_proj GEMM again
  "eg"          the gate / up projections e, g (the 2 x 14336-wide tensors: 57 % of a layer's activations)
                                                                     else: norm2 -> gate/up GEMM again
  never needed  the SwiGLU output h and the down projection: the activation backward rebuilds h from (e, g) in
                place and nothing in the backward reads the layer's own output -- a reentrant checkpoint
                recomputes both anyway (27 % of a layer's GEMM flops + two streaming passes).

Recomputed tensors come from the same deterministic kernels on the same inputs, so every policy gives BITWISE the
gradients of the keep-everything path (tests/test_gpu_model.py).

Policies: "min" = {} (memory of plain checkpointing, ~30 % less recompute), "attn" = {qkv, h1} (re-runs only
norm2 + gate/up), "all" = everything (no recompute; equals use_gradient_checkpointing=False). The bare spelling
"unsloth" = "unsloth:auto": `all` for as many layers as the free HBM holds, `attn` for the rest (auto_schedule).
"""
import torch

from .. import nf4 as _nf4
from ..kernels import attention as _flash
from ..kernels.fast_lora import (
    get_lora_parameters, mlp_backward, mlp_forward, mlp_gate_up_forward, qkv_backward, qkv_forward, w_backward,
    w_forward,
)
from ..kernels.rms_layernorm import add_rms_fwd, rms_bwd_, rms_fwd
from ..kernels.utils import lora_linear_forward
from ..kernels.rope_embedding import Fast_MRoPE_Embedding_QK, _launch_qk, _tables
from ..kernels.swiglu import swiglu_DWf_DW_dfg_kernel, swiglu_fg_kernel

POLICIES = {"min": frozenset(), "attn": frozenset({"qkv", "h1"}), "all": frozenset({"x1", "qkv", "h1", "x2", "eg"})}
_PROJ = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def resolve_policy(policy):
    if isinstance(policy, str):
        return POLICIES[policy]
    return frozenset(policy)


AUTO = "auto"


def auto_schedule(n_layers, tokens, hidden, inter, qkv_cols, elsize, free_bytes, vocab=0, headroom=0.15):
    """"unsloth" (= "unsloth:auto"): the least-recompute schedule `all*k,attn` that fits. The reference's "unsloth" mode is the same kind
    of decision in the other direction (models/_utils.py:360-386 + unsloth_zoo: offload layer inputs to host RAM when VRAM is
    short); with 288 GB of HBM the question is how many layers can simply KEEP everything. Pure arithmetic (tested on the CPU):
      a layer under "attn" keeps  layer input + Q|K|V + attention output (+ fp32 LSE) + post-attention residual,
      a layer under "all" keeps   that + both normed inputs + e and g ([tokens, 2 * inter]);
    `free_bytes` = HBM that the step may use (driver-free + the allocator's cached blocks); the transient working set of one
    layer (e, g, h, their gradients, decoded weight scratch) and of the loss (two logits chunks of <= 4096 rows) is set aside
    first, `headroom` of the rest stays unused. Returns [(k, all), (None, attn)], or the uniform policy when k is 0 / n_layers."""
    per_tok_attn = (hidden + qkv_cols + hidden + hidden) * elsize + 4 * (qkv_cols // 128 + 1)
    per_tok_all_extra = (2 * hidden + 2 * inter) * elsize
    transient = tokens * (6 * inter + 4 * hidden + 2 * qkv_cols) * elsize + 2 * min(tokens, 4096) * max(vocab, 1) * elsize \
        + 3 * (hidden * inter) * elsize
    budget = (free_bytes - transient - n_layers * tokens * per_tok_attn) * (1.0 - headroom)
    k = int(budget // max(1, tokens * per_tok_all_extra))
    k = max(0, min(n_layers, k))
    if k == 0:
        return POLICIES["attn"]
    if k == n_layers:
        return POLICIES["all"]
    return [(k, POLICIES["all"]), (None, POLICIES["attn"])]


def mirrors_fit(n_layers, tokens, hidden, inter, qkv_cols, elsize, free_bytes, vocab=0, headroom=0.15, have_mirrors=False):
    """The second half of the fit-to-memory decision: may the NF4 projections keep decoded 16-bit mirrors (nf4.RESIDENT_MODE
    "auto")? Yes when every layer keeps everything (auto_schedule) and what is left of `free_bytes` after that, headroom
    set aside, holds the mirrors (2 B per projection parameter) once over with as much again to spare. `have_mirrors`: they
    are allocated already (free_bytes no longer contains them): the question is then whether "all" still fits without
    giving them back. Same arithmetic as auto_schedule, same CPU test."""
    per_tok_attn = (hidden + qkv_cols + hidden + hidden) * elsize + 4 * (qkv_cols // 128 + 1)
    per_tok_all_extra = (2 * hidden + 2 * inter) * elsize
    transient = tokens * (6 * inter + 4 * hidden + 2 * qkv_cols) * elsize + 2 * min(tokens, 4096) * max(vocab, 1) * elsize \
        + 3 * (hidden * inter) * elsize
    left = (free_bytes - transient - n_layers * tokens * (per_tok_attn + per_tok_all_extra)) * (1.0 - headroom)
    mirrors = n_layers * (hidden * qkv_cols + hidden * hidden + 3 * hidden * inter) * elsize
    return left >= 0 if have_mirrors else left >= 2 * mirrors


def free_hbm_bytes(dev):
    """HBM this step may still take: what the driver reports free + the blocks torch's allocator holds but has not handed
    out. UNSLOTH_AMD_GC_FREE_GB caps it (a share of a GPU that other jobs use; the capped operating point of bench.py)."""
    import os
    free, _total = torch.cuda.mem_get_info(dev)
    free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    cap = os.environ.get("UNSLOTH_AMD_GC_FREE_GB", "")
    if cap:
        free = min(free, int(float(cap) * (1 << 30)))
    return free


def auto_policy(model, hidden_states):
    """auto_schedule for this call: widths from the config, tokens from the batch, free HBM from the driver + torch's cache."""
    cfg = model.config
    dev = hidden_states.device
    free = free_hbm_bytes(dev)
    head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
    qkv_cols = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * head_dim
    tokens = hidden_states.shape[0] * hidden_states.shape[1]
    key = (tokens, free >> 30)
    hit = getattr(model, "_uamd_auto_policy", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    args = (len(model.layers), tokens, cfg.hidden_size, cfg.intermediate_size, qkv_cols, hidden_states.element_size(), free)
    vocab = getattr(cfg, "vocab_size", 0)
    pol = auto_schedule(*args, vocab=vocab)
    from .. import nf4 as _nf4
    if _nf4.RESIDENT_MODE == "auto":
        # decoded mirrors of the NF4 weights ride on the same decision (nf4.py): on when there is HBM to spare, off when not
        # -- of THIS model's projections only (the switch sits on its quant states): another NF4 model in the process (a
        # reference / policy model, an engine) decides for itself, from the memory that is free when ITS first step runs
        auto_on = getattr(model, "_uamd_mirrors_auto", False)
        callers = _nf4.mirrors_on(model) and not auto_on        # switched on by the caller (model=m or process-wide): not ours
        want = pol == POLICIES["all"] and mirrors_fit(*args, vocab=vocab, have_mirrors=auto_on and _nf4.resident_count(model) > 0)
        if want and not auto_on and not callers:
            _nf4.set_resident(True, auto=True, model=model)
        elif not want and auto_on:                    # (mirrors a caller switched on with nf4.set_resident(True) are the caller's)
            _nf4.set_resident(False, model=model)
            torch.cuda.empty_cache()
    # the step decode (nf4.STEP_DECODE_MODE): decoded weights kept from a layer's forward to its backward when nothing is
    # recomputed and one decoded copy of the projections fits beside everything else
    if _nf4.STEP_DECODE_MODE == "auto":
        model._uamd_step_decode = pol == POLICIES["all"] and not _nf4.mirrors_on(model) and mirrors_fit(*args, vocab=vocab)
    else:
        model._uamd_step_decode = _nf4.STEP_DECODE_MODE == "1" and pol == POLICIES["all"]
    model._uamd_auto_policy = (key, pol)
    return pol


def resolve_policy_spec(spec):
    """"attn" | "qkv+h1" | "all*3,attn": one policy, or a comma-separated schedule over the layers in order -- `name*count`
    covers `count` layers, a segment without a count covers all remaining ones (speed/memory dial between two policies:
    every layer moved from "attn" to "all" trades ~0.45 GB at 8192 tokens of Llama-3-8B for one gate/up GEMM less in the
    backward). Returns a frozenset (uniform) or a list of (count or None, frozenset)."""
    def one(name):
        name = name.strip()
        return resolve_policy(name.split("+") if "+" in name else name)
    if spec.strip() == AUTO:
        return AUTO                     # decided per call from the batch size and the free HBM (auto_policy)
    if "," not in spec and "*" not in spec:
        return one(spec)
    out = []
    for seg in spec.split(","):
        name, _, cnt = seg.partition("*")
        out.append((int(cnt) if cnt else None, one(name)))
    return out


def policy_for_layer(policy, index):
    """The policy of decoder layer `index` under a uniform policy or a schedule (resolve_policy_spec)."""
    if not isinstance(policy, list):
        return policy
    i = 0
    for cnt, pol in policy:
        if cnt is None or index < i + cnt:
            return pol
        i += cnt
    return policy[-1][1]


def layer_supported(layer, hidden, attention_mask):
    """The whole-layer Function covers what the fused hooks cover: LoRA (or plain frozen) projections without bias /
    dropout / DoRA, SwiGLU MLP, flash attention at a head_dim that is a multiple of 8 up to 128, no key-padding mask, 16-bit activations."""
    from ..kernels.fast_lora import apply_lora_mlp_swiglu, apply_lora_o, apply_lora_qkv
    attn, mlp = layer.self_attn, layer.mlp
    if attention_mask is not None or hidden.dtype not in (torch.bfloat16, torch.float16):
        return False
    if getattr(attn, "apply_qkv", None) is not apply_lora_qkv or getattr(attn, "apply_o", None) is not apply_lora_o:
        return False
    if getattr(getattr(mlp, "forward", None), "__func__", None) is not apply_lora_mlp_swiglu:
        return False
    for m in [getattr(attn, n) for n in _PROJ[:4]] + [getattr(mlp, n) for n in _PROJ[4:]]:
        b = getattr(getattr(m, "base_layer", m), "bias", None)
        if b is not None and (b.requires_grad or m in (mlp.gate_proj, mlp.up_proj, mlp.down_proj)):
            return False
    cfg = getattr(attn, "config", None)
    groups = (cfg.num_attention_heads // cfg.num_key_value_heads) if cfg is not None else 1
    # (every group size 1 .. 8 is native to csrc/attention.hip)
    # (head dims below 128 -- TinyLlama / Llama-3.2-1B: 64 -- are native to the attention kernels since round 6)
    return (attn.head_dim <= 128 and attn.head_dim % 8 == 0 and groups <= 8
            and getattr(layer.input_layernorm, "weight", None) is not None)


def _eps(norm):
    return norm.variance_epsilon if hasattr(norm, "variance_epsilon") else norm.eps


class _Static:
    """Non-tensor context of one layer call (weights, norm parameters, rope tables, band, shapes)."""
    __slots__ = ("projs", "biases", "w1", "w2", "eps1", "eps2", "cos", "sin", "idx", "band", "n_heads", "n_kv", "head_dim",
                 "scale", "keep", "shape", "step_decode")


def _rope(st, q4, k4, backward):
    """RoPE in place on the [B, T, H, D] projection outputs: per-token indices (or none), or -- `st.idx` a
    (positions3 int32 [3, B*T], (s_t, s_h, s_w)) pair -- the multimodal variant of the Qwen2-VL text tower."""
    q, k = q4.transpose(1, 2), k4.transpose(1, 2)
    if isinstance(st.idx, tuple):
        pos3, sec = st.idx
        Fast_MRoPE_Embedding_QK._run(q, k, st.cos, st.sin, pos3, int(sec[0]), int(sec[1]), backward)
    else:
        _launch_qk(q, k, st.cos, st.sin, st.idx, backward)


def _with_bias(projs, biases, idx):
    return [projs[i] + (biases[i],) for i in idx]


class DecoderLayerFunction(torch.autograd.Function):
    """(residual', delta') = layer(residual, delta): hidden = residual + delta is formed inside the first norm,
    residual' = hidden + attention block, delta' = MLP output (its add is fused into the NEXT layer's first norm,
    like models/llama.py LlamaDecoderLayer_fused_residual_forward). `delta` may be None (first layer).
    `lora` = the 14 LoRA factors (A, B of q, k, v, o, gate, up, down; None where a projection has no adapter): they
    are inputs so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, st, residual, delta, *lora):
        keep = st.keep
        shape = residual.shape
        st.shape = shape
        projs = st.projs
        if st.step_decode:
            _nf4.step_keep([p_[1] for p_ in projs])
        # ---- attention block
        if delta is None:
            h0 = residual.reshape(-1, shape[-1])
            x1, r1 = rms_fwd(h0, st.w1, st.eps1)
        else:
            h0, x1, r1 = add_rms_fwd(delta, residual, st.w1, st.eps1)
        Q, K, V, xa_qkv = qkv_forward(x1, *_with_bias(projs, st.biases, (0, 1, 2)))
        B_, T_ = shape[0], shape[1]
        q4 = Q.view(B_, T_, st.n_heads, st.head_dim)
        k4 = K.view(B_, T_, st.n_kv, st.head_dim)
        v4 = V.view(B_, T_, st.n_kv, st.head_dim)
        _rope(st, q4, k4, False)                                                                 # RoPE in place
        O, lse = _flash.attn_forward(q4, k4, v4, st.scale, st.band)
        attn = O.view(-1, st.n_heads * st.head_dim)
        o, xa_o = w_forward(attn, projs[3] + (st.biases[3],))
        h1, x2, r2 = add_rms_fwd(o, h0, st.w2, st.eps2)
        del o
        # ---- MLP block
        out, e, g, xa_mlp = mlp_forward(x2, projs[4], projs[5], projs[6], swiglu_fg_kernel)
        saved = [h0, r1, r2, *xa_qkv, xa_o, *xa_mlp]
        ctx.n_fixed = len(saved)
        if "x1" in keep:
            saved.append(x1)
        if "qkv" in keep:
            saved += [Q, K, V, O, lse]
        if "h1" in keep:
            saved.append(h1)
        if "x2" in keep:
            saved.append(x2)
        if "eg" in keep:
            saved += [e, g]
        ctx.save_for_backward(*saved, *[t for t in lora if t is not None])
        ctx.n_saved = len(saved)
        ctx.lora_mask = [t is not None for t in lora]
        ctx.st = st
        ctx.first = delta is None
        ctx.set_materialize_grads(False)
        return h1.view(shape), out.view(shape)

    @staticmethod
    def backward(ctx, d_h1, d_out):
        st = ctx.st
        keep = st.keep
        saved = ctx.saved_tensors
        h0, r1, r2, xa_q, xa_k, xa_v, xa_o, xa_g, xa_u, xa_d = saved[:ctx.n_fixed]
        rest = list(saved[ctx.n_fixed:ctx.n_saved])
        lora_live = list(saved[ctx.n_saved:])
        lora = [lora_live.pop(0) if m else None for m in ctx.lora_mask]
        # projections with the adapters as autograd handed them back (same objects as in the forward)
        projs = [(W, qs, lora[2 * i], lora[2 * i + 1], s) for i, (W, qs, _, _, s) in enumerate(st.projs)]
        shape = st.shape
        B_, T_ = shape[0], shape[1]
        hd, nh, nkv = st.head_dim, st.n_heads, st.n_kv
        x1 = rest.pop(0) if "x1" in keep else None
        if "qkv" in keep:
            Q, K, V, O, lse = rest[:5]
            del rest[:5]
        h1 = rest.pop(0) if "h1" in keep else None
        x2 = rest.pop(0) if "x2" in keep else None
        if "eg" in keep:
            e, g = rest[:2]
        # ---- recompute what the policy did not keep (same kernels, same inputs: bitwise the forward's values)
        with torch.no_grad():
            if x1 is None:
                x1, _ = rms_fwd(h0, st.w1, st.eps1)
            if "qkv" not in keep:
                Q, K, V = lora_linear_forward(x1, _with_bias(projs, st.biases, (0, 1, 2)))
                q4 = Q.view(B_, T_, nh, hd)
                k4 = K.view(B_, T_, nkv, hd)
                _rope(st, q4, k4, False)
                O, lse = _flash.attn_forward(q4, k4, V.view(B_, T_, nkv, hd), st.scale, st.band)
            attn = O.view(-1, nh * hd)
            if h1 is None:
                o, _ = w_forward(attn, projs[3] + (st.biases[3],))
                h1, x2, _ = add_rms_fwd(o, h0, st.w2, st.eps2)
                del o
            elif x2 is None:
                x2, _ = rms_fwd(h1, st.w2, st.eps2)
            if "eg" not in keep:
                e, g = mlp_gate_up_forward(x2, projs[4], projs[5])
            # ---- MLP backward (consumes e, g in place; dX into x2's buffer)
            if d_out is None:
                d_out = torch.zeros(shape, dtype=h0.dtype, device=h0.device)
            # kept e / g / x2 are overwritten by the in-place backward exactly like the reference overwrites its
            # saved tensors (fast_lora.py:157, :193-204): one backward per forward
            dx2, g_mlp = mlp_backward(d_out, x2, e, g, (xa_g, xa_u, xa_d), projs[4], projs[5], projs[6],
                                      swiglu_DWf_DW_dfg_kernel, True)
            del e, g
            # ---- norm2 backward + residual gradient: d h1 = rms'(dx2) + d_h1
            dh1 = rms_bwd_(dx2, h1, st.w2, r2, d_h1)
            # ---- o_proj backward
            d_attn, g_o = w_backward(dh1, attn, xa_o, projs[3])
            # ---- attention backward, RoPE backward (in place), q/k/v backward
            dq, dk, dv = _flash.attn_backward(d_attn.view(B_, T_, nh, hd), Q.view(B_, T_, nh, hd),
                                              K.view(B_, T_, nkv, hd), V.view(B_, T_, nkv, hd),
                                              O.view(B_, T_, nh, hd), lse, st.scale, st.band)
            _rope(st, dq, dk, True)
            dx1, g_qkv = qkv_backward(dq.reshape(-1, nh * hd), dk.reshape(-1, nkv * hd), dv.reshape(-1, nkv * hd), x1,
                                      (xa_q, xa_k, xa_v), projs[0], projs[1], projs[2], True)
            # ---- norm1 backward + the residual path: d h0 = rms'(dx1) + d h1
            need_in = ctx.needs_input_grad[1] or (not ctx.first and ctx.needs_input_grad[2])
            dh0 = rms_bwd_(dx1, h0, st.w1, r1, dh1).view(shape) if need_in else None
            if st.step_decode:
                _nf4.step_release([p_[1] for p_ in st.projs])
        grads = [g_qkv[0], g_qkv[1], g_qkv[2], g_qkv[3], g_qkv[4], g_qkv[5], g_o[0], g_o[1],
                 g_mlp[0], g_mlp[1], g_mlp[2], g_mlp[3], g_mlp[4], g_mlp[5]]
        grads = [gr if m else None for gr, m in zip(grads, ctx.lora_mask)]
        return (None, dh0, None if ctx.first else dh0, *grads)


def decoder_layer_forward(layer, residual, delta, cos, sin, rope_position_ids, band, keep, step_decode=False):
    """(residual', delta') through DecoderLayerFunction. `keep`: a resolved policy (frozenset). `step_decode`: the layer's
    decoded NF4 weights live until its backward (nf4.STEP_DECODE_MODE; only with the keep-everything policy)."""
    attn = layer.self_attn
    cfg = attn.config
    st = _Static()
    mods = [getattr(attn, n) for n in _PROJ[:4]] + [getattr(layer.mlp, n) for n in _PROJ[4:]]
    st.projs = [get_lora_parameters(m) for m in mods]
    # frozen biases of the attention projections (Qwen2: q/k/v) ride in the GEMM epilogue; a TRAINABLE bias needs its
    # own gradient and keeps the per-block Functions (layer_supported)
    st.biases = [getattr(getattr(m, "base_layer", m), "bias", None) for m in mods]
    st.w1, st.w2 = layer.input_layernorm.weight, layer.post_attention_layernorm.weight
    st.eps1, st.eps2 = _eps(layer.input_layernorm), _eps(layer.post_attention_layernorm)
    st.cos, st.sin = _tables(cos, sin)
    if isinstance(rope_position_ids, tuple):          # (positions3 [3, B, T], mrope_section)
        st.idx = (rope_position_ids[0].reshape(3, -1).contiguous(), tuple(rope_position_ids[1]))
        assert sum(st.idx[1]) == attn.head_dim // 2, "mrope_section must cover head_dim / 2 rotary pairs"
    else:
        st.idx = rope_position_ids
    st.band = band
    st.n_heads, st.n_kv, st.head_dim = cfg.num_attention_heads, cfg.num_key_value_heads, attn.head_dim
    st.scale = None
    st.keep = keep
    st.step_decode = bool(step_decode) and keep == POLICIES["all"]
    lora = []
    for (_, _, A, B, _) in st.projs:
        lora += [A, B]
    return DecoderLayerFunction.apply(st, residual, delta, *lora)
