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


def _vl_text_config(cfg):
    """transformers' Qwen2VLTextConfig with the dimensions / rope parameters of the product's Qwen2 tower config."""
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLTextConfig
    d = cfg.to_dict()
    keep = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
            "num_key_value_heads", "hidden_act", "max_position_embeddings", "rms_norm_eps", "tie_word_embeddings")
    kw = {k: d[k] for k in keep if k in d}
    return Qwen2VLTextConfig(**kw, rope_parameters=dict(d.get("rope_parameters") or {}))


def hf_reference_loss_and_lora_grads(fast_model, input_ids, labels, position_ids=None, n_items=None, device="cpu",
                                     return_model=False, loss_fn=None, dtype=torch.float32):
    """Returns (loss fp32, {param_name: grad}) computed by a stock HF model in fp32 -- on the CPU by default; `device`
    = "cuda" runs the same stock-HF fp32 composition on the GPU (torch's own fp32 GEMMs and eager attention: still no
    kernel of the product) so that the BASELINE configurations can be checked at their stated sizes in seconds.
    position_ids [3, B, T] (multimodal RoPE): the backbone is transformers' Qwen2VLTextModel + an lm_head matmul.
    Biases of the base projections (Qwen2's q/k/v) are copied. `loss_fn(logits fp32 [B, T, V]) -> scalar` replaces the
    causal-LM cross entropy (per-token log-prob objectives of the GRPO / DPO path). `dtype` = torch.bfloat16 runs the SAME
    stock composition in bf16 (weights rounded once, activations in bf16, HF's own kernels): not an oracle but a
    YARDSTICK -- the error stock HuggingFace itself has against the fp32 truth at that size, which bounds what any bf16
    implementation can be asked for."""
    from transformers import AutoModelForCausalLM
    base = fast_model.get_base_model() if hasattr(fast_model, "get_base_model") else fast_model
    cfg = copy.deepcopy(base.config)
    cfg.dtype = torch.float32
    cfg._attn_implementation = "eager"
    mrope = position_ids is not None and position_ids.dim() == 3
    dev = torch.device(device)
    with _stock_hf_classes():
        if mrope:
            from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextModel
            tcfg = _vl_text_config(cfg)
            tcfg._attn_implementation = "eager"
            backbone = Qwen2VLTextModel(tcfg).to(torch.float32)
            ref = None
        else:
            ref = AutoModelForCausalLM.from_config(cfg).to(torch.float32)
            backbone = ref.model
            # the product patches LlamaForCausalLM.forward at class level: make sure THIS instance runs stock HF
            ref._unsloth_amd_fast = False
    lm_head_w = base.lm_head.weight.detach().float().cpu()
    with torch.no_grad():
        backbone.embed_tokens.weight.copy_(base.model.embed_tokens.weight.detach().float().cpu())
        backbone.norm.weight.copy_(base.model.norm.weight.detach().float().cpu())
        if ref is not None:
            ref.lm_head.weight.copy_(lm_head_w)
    lora = {}
    eff = {}
    for li, (layer, rlayer) in enumerate(zip(base.model.layers, backbone.layers)):
        with torch.no_grad():
            rlayer.input_layernorm.weight.copy_(layer.input_layernorm.weight.detach().float().cpu())
            rlayer.post_attention_layernorm.weight.copy_(layer.post_attention_layernorm.weight.detach().float().cpu())
        for parent, name in NAMES:
            proj = getattr(getattr(layer, parent), name)
            W, A, B, s = _base_and_lora(proj)
            target = getattr(getattr(rlayer, parent), name)
            Weff = (W + s * B @ A) if A is not None else W
            bias = getattr(getattr(proj, "base_layer", proj), "bias", None)
            with torch.no_grad():
                target.weight.copy_(Weff)
                if bias is not None:
                    target.bias.copy_(bias.detach().float().cpu())
                else:
                    assert target.bias is None
            if A is not None:
                lora[(li, parent, name)] = (A, B, s)
                eff[(li, parent, name)] = target.weight
    for p in backbone.parameters():
        p.requires_grad_(False)
    top = ref if ref is not None else backbone
    top.to(device=dev)
    if dtype != torch.float32:
        # PARAMETERS only: `module.to(bf16)` would also round the rotary `inv_freq` buffers (positions x 2^-9 of phase
        # error: garbage from a few hundred tokens on), which no real bf16 run does (HF keeps them in fp32)
        for p in top.parameters():
            p.data = p.data.to(dtype)
    for k in eff:
        eff[k].requires_grad_(True)
    ids = input_ids.to(dev)
    lab = labels.to(dev)
    pos = None if position_ids is None else position_ids.to(dev).long()
    if mrope:
        h = backbone(input_ids=ids, position_ids=pos, use_cache=False).last_hidden_state
        logits = (h @ lm_head_w.to(device=dev, dtype=dtype).t()).float()
    else:
        for p in ref.lm_head.parameters():
            p.requires_grad_(False)
        logits = ref(input_ids=ids, position_ids=pos, use_cache=False).logits.float()
    if loss_fn is not None:
        loss = loss_fn(logits)
    else:
        shift = R.shift_labels(lab)
        V = logits.shape[-1]
        n = torch.count_nonzero(shift != -100) if n_items is None else n_items
        loss = torch.nn.functional.cross_entropy(logits.view(-1, V), shift.view(-1), ignore_index=-100,
                                                 reduction="sum") / n
    grads = {}
    if eff:
        keys = list(eff)
        dWs = torch.autograd.grad(loss, [eff[k] for k in keys])
        for k, dW in zip(keys, dWs):
            A, B, s = lora[k]
            li, parent, name = k
            dW = dW.float().cpu()
            grads[f"layers.{li}.{parent}.{name}.lora_A"] = s * B.t() @ dW
            grads[f"layers.{li}.{parent}.{name}.lora_B"] = s * dW @ A.t()
    if return_model:
        return loss.detach().cpu(), grads, (ref if ref is not None else backbone)
    return loss.detach().cpu(), grads


def hf_reference_loss_and_all_grads(fast_model, input_ids, labels, position_ids=None, n_items=None, device="cpu",
                                    dtype=torch.float32):
    """Full fine-tuning (BASELINE config 3): (loss, {parameter name: gradient}) of a stock HF model holding the SAME
    16-bit weights cast to fp32, every parameter trainable -- embeddings, norms, projections, lm_head -- HF's own forward,
    torch autograd, no kernel of the product. `dtype` = bf16: the yardstick (see hf_reference_loss_and_lora_grads)."""
    from transformers import AutoModelForCausalLM
    cfg = copy.deepcopy(fast_model.config)
    cfg.dtype = torch.float32
    cfg._attn_implementation = "eager"
    with _stock_hf_classes():
        ref = AutoModelForCausalLM.from_config(cfg).to(torch.float32)
    ref._unsloth_amd_fast = False
    sd = {k: v.detach().float().cpu() for k, v in fast_model.state_dict().items()}
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "inv_freq" not in k], missing
    ref.to(torch.device(device))
    if dtype != torch.float32:
        for p in ref.parameters():
            p.data = p.data.to(dtype)
    dev = torch.device(device)
    ids, lab = input_ids.to(dev), labels.to(dev)
    pos = None if position_ids is None else position_ids.to(dev).long()
    logits = ref(input_ids=ids, position_ids=pos, use_cache=False).logits.float()
    shift = R.shift_labels(lab)
    V = logits.shape[-1]
    n = torch.count_nonzero(shift != -100) if n_items is None else n_items
    loss = torch.nn.functional.cross_entropy(logits.view(-1, V), shift.view(-1), ignore_index=-100, reduction="sum") / n
    loss.backward()
    return loss.detach().cpu(), {k: p.grad.detach().float().cpu() for k, p in ref.named_parameters() if p.grad is not None}


def hf_vl_reference_loss_and_lora_grads(fast_vl, input_ids, attention_mask, pixel_values, image_grid_thw, labels,
                                        device="cpu", dtype=torch.float32):
    """BASELINE config 4 end to end (pixel_values -> patch-embed -> ViT -> merger -> scatter -> multimodal RoPE -> language
    tower -> loss): (loss, {name: grad}) of transformers' own Qwen2VLForConditionalGeneration in fp32 holding the product's
    weights -- NF4 decoded by the oracle, LoRA merged (W + s B A) on BOTH towers -- HF's forward, torch autograd. Gradient
    names: "visual.<module>.lora_A|B" and "language.layers.<i>.<block>.<proj>.lora_A|B". `dtype` = bf16: the yardstick."""
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLForConditionalGeneration
    from unsloth_amd import lora as _l
    cfg = copy.deepcopy(fast_vl.config)
    cfg.dtype = torch.float32
    for c in (cfg, getattr(cfg, "text_config", None), getattr(cfg, "vision_config", None)):
        if c is not None:
            c._attn_implementation = "eager"
    with _stock_hf_classes():
        hf = Qwen2VLForConditionalGeneration(cfg).float()
    hf_mods = dict(hf.named_modules())
    lang = fast_vl.language.get_base_model() if hasattr(fast_vl.language, "get_base_model") else fast_vl.language
    lora, eff = {}, {}

    def place(fast_root, hf_prefix, tag):
        for name, mod in fast_root.named_modules():
            is_lora = isinstance(mod, _l.LoraLayer)
            base = getattr(mod, "base_layer", mod)
            if not (is_lora or (hasattr(base, "weight") and getattr(base.weight, "quant_state", None) is not None)
                    or type(mod) is torch.nn.Linear):
                continue
            if ".base_layer" in name or name.endswith("base_layer") or ".lora_" in "." + name:
                continue
            tgt = hf_mods[hf_prefix + name]
            W, A, B, s = _base_and_lora(mod)
            with torch.no_grad():
                tgt.weight.copy_((W + s * B @ A) if A is not None else W)
                bias = getattr(base, "bias", None)
                if bias is not None:
                    tgt.bias.copy_(bias.detach().float().cpu())
            if A is not None:
                lora[tag + name] = (A, B, s)
                eff[tag + name] = tgt.weight

    place(fast_vl.visual, "model.visual.", "visual.")
    place(lang.model, "model.language_model.", "language.")
    with torch.no_grad():
        hf.lm_head.weight.copy_(lang.lm_head.weight.detach().float().cpu())
        # everything that is not a linear layer: norms, embeddings, the patch-embed convolution
        hf_params = dict(hf.named_parameters())
        for root, prefix in ((fast_vl.visual, "model.visual."), (lang.model, "model.language_model.")):
            lin = {n for n, m in root.named_modules() if isinstance(m, (_l.LoraLayer, torch.nn.Linear)) or
                   getattr(getattr(m, "weight", None), "quant_state", None) is not None}
            for n, p in root.named_parameters():
                owner = n.rsplit(".", 1)[0]
                if any(owner == l or owner.startswith(l + ".") for l in lin) or ".lora_" in "." + n:
                    continue
                hf_params[prefix + n].copy_(p.detach().float().cpu())
    for p in hf.parameters():
        p.requires_grad_(False)
    dev = torch.device(device)
    hf.to(dev)
    if dtype != torch.float32:
        for p in hf.parameters():
            p.data = p.data.to(dtype)
    for k in eff:
        eff[k].requires_grad_(True)
    ids = input_ids.to(dev)
    # (the patch-embed Conv3d through torch's own convolution, not MIOpen: its first call on a fresh GPU box spends minutes in
    # kernel search / compilation per dtype)
    with torch.backends.cudnn.flags(enabled=False):
        out = hf(input_ids=ids, attention_mask=None if attention_mask is None else attention_mask.to(dev),
                 pixel_values=pixel_values.to(dev).to(dtype), image_grid_thw=image_grid_thw.to(dev),
                 mm_token_type_ids=(ids == cfg.image_token_id).int())
    logits = out.logits.float()
    lab = labels.to(dev)
    loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), lab[:, 1:].reshape(-1),
                                             ignore_index=-100)
    keys = list(eff)
    dWs = torch.autograd.grad(loss, [eff[k] for k in keys])
    grads = {}
    for k, dW in zip(keys, dWs):
        A, B, s = lora[k]
        dW = dW.float().cpu()
        grads[k + ".lora_A"] = s * B.t() @ dW
        grads[k + ".lora_B"] = s * dW @ A.t()
    return loss.detach().cpu(), grads
