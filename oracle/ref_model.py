"""CPU ORACLE for the whole composition -- TEST INFRASTRUCTURE ONLY.

Rebuilds, on the CPU in fp32, a stock HuggingFace causal LM whose weights are the (oracle-)dequantised
NF4 base weights with the LoRA update merged in (W + s * B @ A, save.py:622-650 `_merge_lora` semantics),
and runs HF's own forward + loss. That is an implementation-independent statement of what
FastLanguageModel's fused path must compute: HF transformers + PEFT arithmetic, no custom kernels.
LoRA gradients follow from dL/dW_eff:  dA = s * B^T @ dW,  dB = s * dW @ A^T.
"""
import copy
from contextlib import contextmanager

import torch

from . import ref_ops as R


def _base_and_lora(proj):
    base = getattr(proj, "base_layer", proj)
    W = base.weight
    qs = getattr(W, "quant_state", None)
    Wd = R.nf4_dequantize_state(W.detach().cpu(), _cpu_state(qs)).float() if qs is not None else W.detach().float().cpu()
    if hasattr(proj, "lora_A"):
        ad = proj.active_adapters[0]
        return Wd, proj.lora_A[ad].weight.detach().float().cpu(), proj.lora_B[ad].weight.detach().float().cpu(), proj.scaling[ad]
    return Wd, None, None, None


def _cpu_state(qs):
    qs = copy.copy(qs)
    qs.absmax = qs.absmax.cpu()
    qs.code = qs.code.cpu() if qs.code is not None else None
    if qs.state2 is not None:
        qs.state2 = copy.copy(qs.state2)
        qs.state2.absmax = qs.state2.absmax.cpu()
        qs.state2.code = qs.state2.code.cpu()
        qs.offset = qs.offset.cpu()
    return qs


@contextmanager
def _stock_hf_classes():
    """The product's pre_patch() swaps transformers' LlamaRMSNorm class for its fast subclass
    (as unsloth/kernels/rms_layernorm.py:277-286 does). The oracle must be built from the STOCK class."""
    import transformers.models.llama.modeling_llama as m
    cur = m.LlamaRMSNorm
    stock = cur
    while stock.__name__ != "LlamaRMSNorm" and len(stock.__mro__) > 1:
        stock = stock.__mro__[1]
    m.LlamaRMSNorm = stock
    try:
        yield
    finally:
        m.LlamaRMSNorm = cur


NAMES = (("self_attn", "q_proj"), ("self_attn", "k_proj"), ("self_attn", "v_proj"), ("self_attn", "o_proj"),
         ("mlp", "gate_proj"), ("mlp", "up_proj"), ("mlp", "down_proj"))


def hf_reference_loss_and_lora_grads(fast_model, input_ids, labels, position_ids=None, n_items=None):
    """Returns (loss fp32, {param_name: grad}) computed by a stock HF model on the CPU."""
    from transformers import AutoModelForCausalLM
    base = fast_model.get_base_model() if hasattr(fast_model, "get_base_model") else fast_model
    cfg = copy.deepcopy(base.config)
    cfg.dtype = torch.float32
    cfg._attn_implementation = "eager"
    with _stock_hf_classes():
        ref = AutoModelForCausalLM.from_config(cfg).to(torch.float32)
    # the product patches LlamaForCausalLM.forward at class level: make sure THIS instance runs stock HF
    ref._unsloth_amd_fast = False
    with torch.no_grad():
        ref.model.embed_tokens.weight.copy_(base.model.embed_tokens.weight.detach().float().cpu())
        ref.lm_head.weight.copy_(base.lm_head.weight.detach().float().cpu())
        ref.model.norm.weight.copy_(base.model.norm.weight.detach().float().cpu())
    lora = {}
    eff = {}
    for li, (layer, rlayer) in enumerate(zip(base.model.layers, ref.model.layers)):
        with torch.no_grad():
            rlayer.input_layernorm.weight.copy_(layer.input_layernorm.weight.detach().float().cpu())
            rlayer.post_attention_layernorm.weight.copy_(layer.post_attention_layernorm.weight.detach().float().cpu())
        for parent, name in NAMES:
            proj = getattr(getattr(layer, parent), name)
            W, A, B, s = _base_and_lora(proj)
            target = getattr(getattr(rlayer, parent), name)
            Weff = (W + s * B @ A) if A is not None else W
            with torch.no_grad():
                target.weight.copy_(Weff)
            target.weight.requires_grad_(A is not None)
            if A is not None:
                lora[(li, parent, name)] = (A, B, s)
                eff[(li, parent, name)] = target.weight
    ids = input_ids.cpu()
    lab = labels.cpu()
    out = ref(input_ids=ids, position_ids=None if position_ids is None else position_ids.cpu().long(), use_cache=False)
    logits = out.logits.float()
    shift = R.shift_labels(lab)
    V = logits.shape[-1]
    n = torch.count_nonzero(shift != -100) if n_items is None else n_items
    loss = torch.nn.functional.cross_entropy(logits.view(-1, V), shift.view(-1), ignore_index=-100, reduction="sum") / n
    grads = {}
    if eff:
        keys = list(eff)
        dWs = torch.autograd.grad(loss, [eff[k] for k in keys])
        for k, dW in zip(keys, dWs):
            A, B, s = lora[k]
            li, parent, name = k
            grads[f"layers.{li}.{parent}.{name}.lora_A"] = s * B.t() @ dW
            grads[f"layers.{li}.{parent}.{name}.lora_B"] = s * dW @ A.t()
    return loss.detach(), grads
