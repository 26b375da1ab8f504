"""Minimal PEFT-compatible LoRA wrapper, used only when the `peft` package is absent (it is not in
this image). It exposes exactly the attributes the reference's plug-in contract reads in
`get_lora_parameters` (unsloth/kernels/utils.py:335-397):
    proj.base_layer, proj.lora_A[adapter].weight, proj.lora_B[adapter].weight, proj.scaling[adapter],
    proj.disable_adapters, proj.merged, proj.active_adapters
and keeps PEFT's parameter names (`...lora_A.default.weight`, `...lora_B.default.weight`) so adapter
checkpoints stay interchangeable (SURVEY 5 "Checkpoint / resume").
"""
import math
import os
from contextlib import contextmanager
from dataclasses import dataclass, field
from typing import List, Optional, Union

import torch
import torch.nn as nn

try:  # the real thing wins when available
    import peft as _peft  # noqa: F401
    HAS_PEFT = True
except Exception:
    HAS_PEFT = False


@dataclass
class LoraConfig:
    r: int = 16
    lora_alpha: int = 16
    target_modules: Union[List[str], str, None] = None
    lora_dropout: float = 0.0
    bias: str = "none"
    use_rslora: bool = False
    modules_to_save: Optional[List[str]] = None
    init_lora_weights: Union[bool, str] = True
    layers_to_transform: Optional[List[int]] = None
    task_type: str = "CAUSAL_LM"
    peft_type: str = "LORA"
    base_model_name_or_path: Optional[str] = None
    loftq_config: dict = field(default_factory=dict)

    def to_dict(self):
        return dict(self.__dict__)


class LoraLayer(nn.Module):
    """LoRA around a frozen linear layer (dense nn.Linear or nf4.Linear4bit)."""

    def __init__(self, base_layer, adapter_name, r, lora_alpha, lora_dropout, use_rslora, init_lora_weights):
        super().__init__()
        self.base_layer = base_layer
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.r, self.lora_alpha, self.scaling, self.use_dora = {}, {}, {}, {}
        self.lora_A, self.lora_B, self.lora_dropout = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        self.lora_magnitude_vector = nn.ModuleDict()
        self._disable_adapters = False
        self.merged_adapters = []
        self._active_adapter = [adapter_name]
        self.update_layer(adapter_name, r, lora_alpha, lora_dropout, use_rslora, init_lora_weights)

    def update_layer(self, name, r, lora_alpha, lora_dropout, use_rslora, init_lora_weights):
        if r <= 0:
            raise ValueError(f"`r` should be a positive integer value but the value passed is {r}")
        dev = self.base_layer.weight.device
        self.r[name], self.lora_alpha[name] = r, lora_alpha
        self.lora_dropout[name] = nn.Dropout(lora_dropout) if lora_dropout > 0 else nn.Identity()
        # LoRA params are fp32 (prepare_model_for_training upcasts them)
        self.lora_A[name] = nn.Linear(self.in_features, r, bias=False, device=dev, dtype=torch.float32)
        self.lora_B[name] = nn.Linear(r, self.out_features, bias=False, device=dev, dtype=torch.float32)
        self.scaling[name] = lora_alpha / math.sqrt(r) if use_rslora else lora_alpha / r
        self.use_dora[name] = False
        if init_lora_weights:
            nn.init.kaiming_uniform_(self.lora_A[name].weight, a=math.sqrt(5))   # PEFT default
            nn.init.zeros_(self.lora_B[name].weight)
        for p in self.base_layer.parameters():
            p.requires_grad_(False)

    # --- attributes read by get_lora_parameters ---------------------------------------------
    @property
    def disable_adapters(self):
        return self._disable_adapters

    @property
    def merged(self):
        return bool(self.merged_adapters)

    @property
    def active_adapters(self):
        return self._active_adapter

    @property
    def active_adapter(self):
        return self._active_adapter[0]

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    def get_base_layer(self):
        return self.base_layer

    def forward(self, x):
        """Generic (un-fused) path: used by decode / anything that bypasses the fast hooks."""
        from .kernels.fast_lora import LoRA_W
        from .kernels.utils import get_lora_parameters_bias
        W, q, A, B, s, bias = get_lora_parameters_bias(self)
        drop = self.lora_dropout[self._active_adapter[0]]
        if isinstance(drop, nn.Identity) or not self.training:
            if bias is not None:                     # (the bias rides in the GEMM epilogue: the ViT's qkv / proj / fc1 / fc2)
                return LoRA_W.apply(x, W, q, A, B, s, bias)
            return LoRA_W.apply(x, W, q, A, B, s)
        out = LoRA_W.apply(x, W, q, None, None, None)
        out = out + (drop(x).to(A.dtype) @ A.t() @ B.t()).to(out.dtype) * s
        return out if bias is None else out + bias


class LoraModel(nn.Module):
    def __init__(self, model, config, adapter_name="default"):
        super().__init__()
        self.model = model
        self.peft_config = {adapter_name: config}
        if config.modules_to_save:
            # PEFT would clone embed_tokens / lm_head into trainable copies; this stand-in does not, and silently
            # leaving them frozen would train something else than what was asked for
            raise NotImplementedError(f"modules_to_save={config.modules_to_save!r}: trainable copies of whole modules "
                                      "are not implemented by this PEFT stand-in (install `peft` to get them)")
        targets = config.target_modules
        if isinstance(targets, str):
            targets = [targets]
        layers = config.layers_to_transform
        for name, module in list(model.named_modules()):
            leaf = name.rsplit(".", 1)[-1]
            if leaf not in targets or isinstance(module, LoraLayer):
                continue
            if not hasattr(module, "in_features"):
                continue
            if layers is not None:
                idx = [int(p) for p in name.split(".") if p.isdigit()]
                if not idx or idx[0] not in layers:
                    continue
            parent = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
            setattr(parent, leaf, LoraLayer(module, adapter_name, config.r, config.lora_alpha,
                                            config.lora_dropout, config.use_rslora, config.init_lora_weights))
        for n, p in model.named_parameters():
            p.requires_grad_("lora_" in n)

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)


class PeftModelForCausalLM(nn.Module):
    """The slice of peft.PeftModelForCausalLM the reference touches."""

    def __init__(self, model, peft_config, adapter_name="default"):
        super().__init__()
        self.base_model = LoraModel(model, peft_config, adapter_name)
        self.peft_config = self.base_model.peft_config
        self.active_adapter = adapter_name
        self.config = getattr(model, "config", None)

    def forward(self, *args, **kwargs):
        return self.base_model(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name == "base_model":
                raise
            return getattr(self.base_model.model, name)

    def get_base_model(self):
        return self.base_model.model

    def get_nb_trainable_parameters(self):
        t = sum(p.numel() for p in self.parameters() if p.requires_grad)
        a = 0
        for p in self.parameters():
            n = p.numel()
            if p.dtype == torch.uint8 and hasattr(p, "quant_state"):
                n *= 2                                    # two 4-bit codes per byte
            a += n
        return t, a

    def print_trainable_parameters(self):
        t, a = self.get_nb_trainable_parameters()
        print(f"trainable params: {t:,d} || all params: {a:,d} || trainable%: {100 * t / a:.4f}")

    @contextmanager
    def disable_adapter(self):
        layers = [m for m in self.modules() if isinstance(m, LoraLayer)]
        try:
            for m in layers:
                m._disable_adapters = True
            yield
        finally:
            for m in layers:
                m._disable_adapters = False

    def lora_state_dict(self):
        return {k: v for k, v in self.state_dict().items() if "lora_" in k}

    def save_pretrained(self, save_directory, **kwargs):
        import json
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        sd = {("base_model.model." + k.split("base_model.model.", 1)[-1]).replace(".default", ""): v.detach().cpu().contiguous()
              for k, v in self.lora_state_dict().items()}
        save_file(sd, os.path.join(save_directory, "adapter_model.safetensors"))
        cfg = {k: v for k, v in self.peft_config[self.active_adapter].to_dict().items()}
        with open(os.path.join(save_directory, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=2, default=str)


def _load_adapter(self, directory, adapter_name=None, is_trainable=True):
    """Reads `adapter_model.safetensors` written by `save_pretrained` (PEFT's key layout: no adapter name in the keys)
    back into the LoRA factors, IN PLACE through `param.copy_` (bumps the version counters the cast caches watch).
    Returns the list of keys that were loaded; raises on missing / unexpected / mis-shaped tensors."""
    from safetensors.torch import load_file
    from .kernels.utils import invalidate_cast_cache
    name = adapter_name or self.active_adapter
    sd = load_file(os.path.join(directory, "adapter_model.safetensors"))
    own = {}
    for k, prm in self.named_parameters():
        if "lora_" in k:
            key = ("base_model.model." + k.split("base_model.model.", 1)[-1]).replace("." + name, "")
            own[key] = prm
    missing, unexpected = sorted(set(own) - set(sd)), sorted(set(sd) - set(own))
    if missing or unexpected:
        raise KeyError(f"adapter checkpoint mismatch: missing {missing[:4]}..., unexpected {unexpected[:4]}...")
    with torch.no_grad():
        for key, prm in own.items():
            t = sd[key]
            if tuple(t.shape) != tuple(prm.shape):
                raise ValueError(f"{key}: checkpoint {tuple(t.shape)} vs model {tuple(prm.shape)}")
            prm.copy_(t.to(device=prm.device, dtype=prm.dtype))
            prm.requires_grad_(bool(is_trainable))
    invalidate_cast_cache()
    return sorted(own)


PeftModelForCausalLM.load_adapter = _load_adapter


def _save_pretrained_merged(self, save_directory, tokenizer=None, save_method="merged_16bit", **kwargs):
    """`model.save_pretrained_merged` of the reference (save.py): "lora" = adapters only, "merged_16bit" = dense
    16-bit safetensors with the adapters folded in, "merged_4bit"/"4bit" is the frozen NF4 base as it sits in HBM."""
    from . import checkpoint as _ckpt
    if save_method == "lora":
        self.save_pretrained(save_directory)
    elif save_method in ("merged_16bit", "16bit"):
        _ckpt.save_pretrained_merged(self, save_directory)
    elif save_method in ("base_4bit", "4bit"):
        _ckpt.save_pretrained_4bit(self, save_directory)
    else:
        raise NotImplementedError(f"save_method={save_method!r}: 'lora', 'merged_16bit', 'base_4bit' (GGUF and "
                                  "re-quantised merged_4bit exports are outside the hot path)")
    if tokenizer is not None:
        tokenizer.save_pretrained(save_directory)


PeftModelForCausalLM.save_pretrained_merged = _save_pretrained_merged


def get_peft_model(model, peft_config, adapter_name="default"):
    return PeftModelForCausalLM(model, peft_config, adapter_name)
