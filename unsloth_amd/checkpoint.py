"""bitsandbytes-4bit checkpoints WITHOUT bitsandbytes: read, write, merge.

What this replaces in the reference:
  * loading `*-bnb-4bit` repos: the reference passes a `BitsAndBytesConfig(nf4, double quant)` to transformers
    (unsloth/models/llama.py:2615-2626) and stamps that config on the model (unsloth/models/loader.py:1022-1053);
    transformers then needs bitsandbytes to deserialise `Params4bit` (quantizers/quantizer_bnb_4bit.py
    `get_weight_conversions`: the per-weight keys below). Neither library exists in this image, so the
    safetensors are parsed here directly into `nf4.Linear4bit` modules -- byte for byte, no re-quantisation;
  * `_merge_lora` (unsloth/save.py:622-650): W = dequant(W) in fp32, W^T += s * A^T B^T, cast back.

On-disk layout of one NF4 linear (bitsandbytes `QuantState.as_dict(packed=True)`, the keys transformers lists):
  <name>.weight                               uint8 [out*in/2, 1]   two codes per byte, high nibble first
  <name>.weight.absmax                        uint8 [out*in/64]     8-bit codes of (absmax - offset) (nested)
  <name>.weight.quant_map                     fp32  [16]            the NF4 code book
  <name>.weight.nested_absmax                 fp32  [out*in/64/256]
  <name>.weight.nested_quant_map              fp32  [256]           dynamic 8-bit map
  <name>.weight.quant_state.bitsandbytes__nf4 uint8 [json bytes]    {quant_type, blocksize, dtype, shape, nested_*}
Everything here is byte / index shuffling on whatever device the tensors live on; the only arithmetic is the merge,
which runs through the HIP dequantiser (GPU only, raises otherwise).
"""
import json
import os

import torch

from . import nf4 as _nf4

QUANT_SUFFIXES = ("absmax", "quant_map", "nested_absmax", "nested_quant_map", "quant_state.bitsandbytes__nf4",
                  "quant_state.bitsandbytes__fp4")
WEIGHTS_NAME, INDEX_NAME = "model.safetensors", "model.safetensors.index.json"


def bnb_quantization_config(compute_dtype):
    """The config block the reference stamps on a 4-bit model (loader.py:1030-1046)."""
    if isinstance(compute_dtype, torch.dtype):
        compute_dtype = str(compute_dtype).replace("torch.", "")
    return {
        "bnb_4bit_compute_dtype": compute_dtype, "bnb_4bit_quant_type": "nf4", "bnb_4bit_use_double_quant": True,
        "llm_int8_enable_fp32_cpu_offload": False, "llm_int8_has_fp16_weight": False,
        "llm_int8_skip_modules": None, "llm_int8_threshold": 6.0, "load_in_4bit": True, "load_in_8bit": False,
        "quant_method": "bitsandbytes",
    }


def is_prequantized(config):
    """True for a config that declares a bitsandbytes 4-bit checkpoint."""
    qc = getattr(config, "quantization_config", None)
    if qc is None:
        return False
    if hasattr(qc, "to_dict"):
        qc = qc.to_dict()
    method = qc.get("quant_method", "bitsandbytes")
    if hasattr(method, "value"):
        method = method.value
    return str(method) == "bitsandbytes" and bool(qc.get("load_in_4bit", False))


def _base(model):
    while hasattr(model, "get_base_model") and model.get_base_model() is not model:
        model = model.get_base_model()
    return model


# ------------------------------------------------------------------------------------------------
# writing
def state_dict_4bit(model):
    """HF-named tensors of the BASE model with every `Linear4bit` expanded into the six bitsandbytes entries.
    LoRA wrappers are looked through (their base layer is what is saved); adapters are not included."""
    from .lora import LoraLayer
    out = {}
    base = _base(model)
    seen_params = set()
    for mod_name, mod in base.named_modules():
        if isinstance(mod, LoraLayer):
            continue
        clean = mod_name.replace(".base_layer", "")
        if isinstance(mod, _nf4.Linear4bit):
            w = mod.weight
            out[clean + ".weight"] = w.data
            for k, v in w.quant_state.as_dict(packed=True).items():
                out[clean + ".weight." + k] = v
            if mod.bias is not None:
                out[clean + ".bias"] = mod.bias.data
            seen_params.update({id(w), id(mod.bias)})
            continue
        for pn, p in mod.named_parameters(recurse=False):
            if id(p) in seen_params or "lora_" in mod_name:
                continue
            seen_params.add(id(p))
            out[(clean + "." if clean else "") + pn] = p.data
    cfg = getattr(base, "config", None)
    if cfg is not None and getattr(cfg, "tie_word_embeddings", False):
        out.pop("lm_head.weight", None)
    return out


def _save_sharded(tensors, directory, max_shard_bytes):
    from safetensors.torch import save_file
    os.makedirs(directory, exist_ok=True)
    shards, cur, cur_bytes = [], {}, 0
    for k, v in tensors.items():
        nb = v.numel() * v.element_size()
        if cur and cur_bytes + nb > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[k] = v.detach().cpu().contiguous()
        cur_bytes += nb
    shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(directory, WEIGHTS_NAME), metadata={"format": "pt"})
        return [WEIGHTS_NAME]
    names, weight_map, total = [], {}, 0
    for i, sh in enumerate(shards):
        fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(directory, fn), metadata={"format": "pt"})
        names.append(fn)
        for k, v in sh.items():
            weight_map[k] = fn
            total += v.numel() * v.element_size()
    with open(os.path.join(directory, INDEX_NAME), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
    return names


def _save_config(model, directory, quantized):
    base = _base(model)
    cfg = getattr(base, "config", None)
    if cfg is None:
        return
    d = cfg.to_dict()
    if quantized:
        d["quantization_config"] = bnb_quantization_config(getattr(cfg, "dtype", None) or torch.bfloat16)
    else:
        d.pop("quantization_config", None)
    with open(os.path.join(directory, "config.json"), "w") as f:
        json.dump(d, f, indent=2, default=str)


def save_pretrained_4bit(model, directory, max_shard_size=5 * 1024 ** 3):
    """Write the frozen base model exactly as it sits in HBM (NF4 bytes + statistics): the `*-bnb-4bit` format."""
    files = _save_sharded(state_dict_4bit(model), directory, max_shard_size)
    _save_config(model, directory, quantized=True)
    return files


# ------------------------------------------------------------------------------------------------
# reading
def checkpoint_files(directory):
    idx = os.path.join(directory, INDEX_NAME)
    if os.path.exists(idx):
        with open(idx) as f:
            return sorted(set(json.load(f)["weight_map"].values()))
    if os.path.exists(os.path.join(directory, WEIGHTS_NAME)):
        return [WEIGHTS_NAME]
    raise FileNotFoundError(f"no {WEIGHTS_NAME} / {INDEX_NAME} under {directory!r}")


def iter_checkpoint_tensors(directory, device="cpu"):
    """(name, tensor) for every tensor of a (possibly sharded) safetensors checkpoint."""
    from safetensors import safe_open
    for fn in checkpoint_files(directory):
        with safe_open(os.path.join(directory, fn), framework="pt", device=str(device)) as f:
            for k in f.keys():
                yield k, f.get_tensor(k)


def _split_quant_key(name):
    for suf in QUANT_SUFFIXES:
        tail = ".weight." + suf
        if name.endswith(tail):
            return name[: -len(tail)], suf
    return None, None


def load_prequantized_(model, directory, device, dtype=None):
    """Fill `model` (an HF module tree built from the checkpoint's config, parameters possibly on the meta
    device) from a bitsandbytes-4bit safetensors directory, IN PLACE. Linear layers that come with quantisation
    statistics become `nf4.Linear4bit` holding the checkpoint's bytes unchanged; everything else is copied
    (cast to `dtype` if given). Returns (missing, unexpected) key lists like `load_state_dict`."""
    device = torch.device(device)
    dense, quant = {}, {}
    for name, t in iter_checkpoint_tensors(directory, "cpu"):
        mod, suf = _split_quant_key(name)
        if mod is not None:
            quant.setdefault(mod, {})[suf] = t
        else:
            dense[name] = t
    if any("bitsandbytes__fp4" in s for q in quant.values() for s in q):
        raise NotImplementedError("fp4 checkpoints: only nf4 is implemented (the reference's 4-bit repos are nf4)")
    modules = dict(model.named_modules())
    unexpected, loaded = [], set()
    for mod_name, side in quant.items():
        lin = modules.get(mod_name)
        wkey = mod_name + ".weight"
        if lin is None or wkey not in dense:
            unexpected.append(wkey)
            continue
        packed = dense.pop(wkey)
        if packed.dtype != torch.uint8:
            raise ValueError(f"{wkey}: quantisation statistics present but the weight is {packed.dtype}")
        qs = _nf4.QuantState.from_dict(side, device)
        if qs.quant_type != "nf4":
            raise NotImplementedError(f"{wkey}: quant_type {qs.quant_type!r}")
        if dtype is not None:
            # bnb-4bit repos are often stamped float16 whatever the training dtype: the stamp only says what the
            # decode WRITES, and the GEMM reads the scratch as the activation dtype. The reference overwrites it
            # with the model dtype after load (models/granite.py:586-596); so do we.
            qs.dtype = dtype
        out_f, in_f = qs.shape
        if packed.numel() * 2 != out_f * in_f:
            raise ValueError(f"{wkey}: {packed.numel()} packed bytes for shape {tuple(qs.shape)}")
        bias = None
        if mod_name + ".bias" in dense:
            bias = torch.nn.Parameter(dense.pop(mod_name + ".bias").to(device=device, dtype=dtype or qs.dtype),
                                      requires_grad=False)
            loaded.add(mod_name + ".bias")
        new = _nf4.Linear4bit(in_f, out_f, packed.reshape(-1, 1).to(device), qs, bias)
        parent_name, _, child = mod_name.rpartition(".")
        setattr(modules[parent_name] if parent_name else model, child, new)
        loaded.add(wkey)
    params = dict(model.named_parameters())
    buffers = dict(model.named_buffers())
    for name, t in dense.items():
        tgt = params.get(name)
        if tgt is None:
            tgt = buffers.get(name)
        if tgt is None:
            unexpected.append(name)
            continue
        val = t.to(device=device, dtype=(dtype if (dtype is not None and t.is_floating_point()) else None))
        if tuple(val.shape) != tuple(tgt.shape):
            raise ValueError(f"{name}: checkpoint {tuple(val.shape)} vs model {tuple(tgt.shape)}")
        mod_name, _, leaf = name.rpartition(".")
        owner = modules[mod_name] if mod_name else model
        if name in params:
            setattr(owner, leaf, torch.nn.Parameter(val, requires_grad=False))
        else:
            owner._buffers[leaf] = val
        loaded.add(name)
    cfg = getattr(model, "config", None)
    if cfg is not None and getattr(cfg, "tie_word_embeddings", False) and "lm_head.weight" not in loaded:
        emb = model.get_input_embeddings()
        head = model.get_output_embeddings()
        if emb is not None and head is not None:
            head.weight = emb.weight
            loaded.add("lm_head.weight")
    # every parameter the module tree still holds must have come from the checkpoint (after `to_empty` nothing is on
    # the meta device any more, so the device cannot be the test: a tensor the checkpoint lacks would otherwise stay
    # uninitialised HBM). Persistent buffers likewise; non-persistent ones (rotary inv_freq) are rebuilt below.
    missing = [n for n, _ in model.named_parameters() if n not in loaded]
    non_persistent = {f"{mn}.{bn}" if mn else bn for mn, m in model.named_modules()
                      for bn in getattr(m, "_non_persistent_buffers_set", ())}
    missing += [n for n, _ in model.named_buffers() if n not in loaded and n not in non_persistent]
    reinit_rotary_buffers_(model, device)
    return missing, unexpected


def load_language_tower_(model, directory):
    """Fill a causal LM built from a VLM's text config (models/loader.py: FastModel) with the language-tower tensors of
    the VLM checkpoint in `directory`: `model.language_model.X` / `language_model.model.X` -> `model.X`, `lm_head.*` as is;
    vision-tower tensors are skipped. Returns the parameter names the checkpoint did not provide."""
    own = dict(model.named_parameters())
    own.update(dict(model.named_buffers()))
    seen = set()
    for name, t in iter_checkpoint_tensors(directory):
        key = name
        for prefix, repl in (("model.language_model.", "model."), ("language_model.model.", "model."),
                             ("language_model.lm_head.", "lm_head.")):
            if key.startswith(prefix):
                key = repl + key[len(prefix):]
                break
        if key in own:
            with torch.no_grad():
                own[key].copy_(t.to(device=own[key].device, dtype=own[key].dtype))
            seen.add(key)
    tied = getattr(model.config, "tie_word_embeddings", False)
    return [n for n, _ in model.named_parameters() if n not in seen and not (tied and n == "lm_head.weight")]


def load_prefixed_(module, directory, prefixes):
    """Fill `module` with the checkpoint tensors whose names start with one of `prefixes` (prefix stripped), e.g. the vision
    tower of a VLM checkpoint ("model.visual." / "visual."). Returns the parameter names the checkpoint did not provide."""
    own = dict(module.named_parameters())
    own.update(dict(module.named_buffers()))
    seen = set()
    for name, t in iter_checkpoint_tensors(directory):
        for prefix in prefixes:
            if name.startswith(prefix) and name[len(prefix):] in own:
                key = name[len(prefix):]
                with torch.no_grad():
                    own[key].copy_(t.to(device=own[key].device, dtype=own[key].dtype))
                seen.add(key)
                break
    return [n for n, _ in module.named_parameters() if n not in seen]


def reinit_rotary_buffers_(model, device):
    """Non-persistent rotary buffers (`inv_freq`, `original_inv_freq`) never travel in a checkpoint; after
    `to_empty` they are uninitialised memory, and HF's own forward (which the decode / generation path uses) reads
    them. Recompute them from the config exactly as the module's constructor does."""
    cfg = getattr(model, "config", None)
    for m in model.modules():
        if "inv_freq" not in getattr(m, "_buffers", {}):
            continue
        init = getattr(m, "rope_init_fn", None)
        mcfg = getattr(m, "config", None) or cfg
        inv = None
        try:
            if init is not None:
                inv, scaling = init(mcfg, device)
                if hasattr(m, "attention_scaling"):
                    m.attention_scaling = scaling
        except Exception:
            inv = None
        if inv is None:
            from .models.llama import compute_inv_freq
            inv, scaling, _ = compute_inv_freq(mcfg)
            if hasattr(m, "attention_scaling"):
                m.attention_scaling = scaling
        inv = inv.to(device=device, dtype=torch.float32)
        m._buffers["inv_freq"] = inv
        if "original_inv_freq" in m._buffers:
            m._buffers["original_inv_freq"] = inv.clone()


# ------------------------------------------------------------------------------------------------
# merging (save.py:622-650)
def merge_lora_weight(layer, name=""):
    """`_merge_lora`: (W_merged [out,in] in the layer's compute dtype, bias). NF4 weights are decoded by the HIP
    kernel; the rank-r update is accumulated in fp32 exactly like the reference:
        W = dequant(W).float().t();  W.addmm_(A.t().float(), B.t().float(), alpha=s);  W = W.t().to(dtype)"""
    from .kernels.utils import get_lora_parameters_bias
    W, quant_state, A, B, s, bias = get_lora_parameters_bias(layer)
    if quant_state is not None:
        dtype = quant_state.dtype
        W = _nf4.dequantize_nf4(W, quant_state)
    else:
        dtype = W.dtype
    W = W.to(torch.float32).t()
    if A is not None:
        W = W.contiguous()
        W.addmm_(A.t().to(torch.float32), B.t().to(torch.float32), alpha=float(s))
        maximum_element = torch.max(W.min().abs(), W.max())
        if not torch.isfinite(maximum_element).item():
            raise ValueError(f"Unsloth: Merge failed.\n{name} has some elements = infinity.")
    return W.t().to(dtype).contiguous(), bias


def merged_state_dict(model):
    """HF-named 16-bit tensors of the model with every adapter folded into its (dequantised) base weight."""
    from .lora import LoraLayer
    base = _base(model)
    out, done = {}, set()
    for mod_name, mod in base.named_modules():
        if isinstance(mod, LoraLayer) or (isinstance(mod, _nf4.Linear4bit) and ".base_layer" not in mod_name):
            W, bias = merge_lora_weight(mod, mod_name)
            out[mod_name + ".weight"] = W
            if bias is not None:
                out[mod_name + ".bias"] = bias.data
            done.add(mod_name)
    for name, p in base.named_parameters():
        mod_name = name.rpartition(".")[0]
        if "lora_" in name or ".base_layer" in name or any(mod_name == d or mod_name.startswith(d + ".") for d in done):
            continue
        out[name] = p.data
    cfg = getattr(base, "config", None)
    if cfg is not None and getattr(cfg, "tie_word_embeddings", False):
        out.pop("lm_head.weight", None)
    return out


def save_pretrained_merged(model, directory, max_shard_size=5 * 1024 ** 3):
    """`save_pretrained_merged(..., save_method="merged_16bit")` of the reference (save.py): dense 16-bit
    safetensors with the adapters merged, loadable by stock transformers."""
    files = _save_sharded(merged_state_dict(model), directory, max_shard_size)
    _save_config(model, directory, quantized=False)
    return files
