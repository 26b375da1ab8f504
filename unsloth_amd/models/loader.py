"""FastLanguageModel: the public surface of the reference (unsloth/models/loader.py:407-1120) over the
MI355X hot path.

    model, tokenizer = FastLanguageModel.from_pretrained(model_name, max_seq_length=2048, dtype=None,
                                                         load_in_4bit=True, ...)
    model = FastLanguageModel.get_peft_model(model, r=16, target_modules=[...], lora_alpha=16, ...)
    model.for_training() / model.for_inference()

Differences forced by this image (no network, no bitsandbytes, no peft): `model_name` must be a local
directory with HF weights, or a `config=` object for random-init (synthetic benchmarks);
4-bit loading quantises on the GPU with our NF4 quantiser (unsloth_amd/nf4.py) or reads a local
bitsandbytes-format checkpoint (`weight.absmax`, `weight.quant_map`, ... keys).
Architectures: dispatch by config.model_type as loader.py:828-890; llama / mistral / qwen2 share the
Llama patcher. Anything else raises (the reference sends it to the unsloth_zoo compiler path: out of scope).
"""
import os

import torch

from .llama import FastLlamaModel, quantize_model_nf4_
from .. import nf4 as _nf4

SUPPORTED = ("llama", "mistral", "qwen2")


def _resolve_dtype(dtype):
    if dtype is None:
        return torch.bfloat16                      # is_bfloat16_supported() on MI355X
    if dtype in (torch.float16, torch.bfloat16):
        return dtype
    raise TypeError("Unsloth: dtype must be None, torch.float16 or torch.bfloat16")


def prepare_device_map():
    """loader_utils.py:91-106: under torchrun every rank owns ONE device (replicas only)."""
    if "LOCAL_RANK" in os.environ:
        return {"": f"cuda:{int(os.environ['LOCAL_RANK'])}"}
    return None


class FastLanguageModel:
    @staticmethod
    def from_pretrained(model_name=None, max_seq_length=2048, dtype=None, load_in_4bit=True,
                        load_in_8bit=False, load_in_16bit=False, full_finetuning=False, token=None,
                        device_map="sequential", rope_scaling=None, fix_tokenizer=True,
                        trust_remote_code=False, use_gradient_checkpointing="unsloth", config=None,
                        device=None, random_state=3407, *args, **kwargs):
        if load_in_8bit or full_finetuning:
            raise NotImplementedError(
                "8-bit / full fine-tuning go through FastModel + the unsloth_zoo compiler in the reference "
                "(loader.py:487-523); outside the QLoRA hot path (SURVEY 8(f4)).")
        dtype = _resolve_dtype(dtype)
        if device is None:
            dm = prepare_device_map()
            device = torch.device(dm[""]) if dm else torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        from transformers import AutoConfig, AutoModelForCausalLM
        tokenizer = None
        if config is None:
            if model_name is None or not os.path.isdir(str(model_name)):
                raise FileNotFoundError(
                    f"Unsloth (MI355X build): {model_name!r} is not a local directory and this environment "
                    "has no network. Pass a local HF checkpoint directory, or config=<PretrainedConfig> for "
                    "a randomly initialised model.")
            config = AutoConfig.from_pretrained(model_name, trust_remote_code=trust_remote_code)
        model_type = getattr(config, "model_type", None)
        if model_type not in SUPPORTED:
            raise NotImplementedError(
                f"model_type={model_type!r}: the hand-kernel path covers {SUPPORTED} (loader.py:828-890).")
        if rope_scaling is not None:
            config.rope_scaling = rope_scaling
        FastLlamaModel.pre_patch()
        prequantized = False
        if model_name is not None and os.path.isdir(str(model_name)):
            from .. import checkpoint as _ckpt
            if _ckpt.is_prequantized(config):
                # `*-bnb-4bit` checkpoint: transformers would need bitsandbytes to deserialise it (llama.py:2615-2626);
                # build the module tree without storage and fill it from the safetensors directly, NF4 bytes unchanged
                import copy
                cfg16 = copy.deepcopy(config)
                if hasattr(cfg16, "quantization_config"):
                    del cfg16.quantization_config
                with torch.device("meta"):
                    model = AutoModelForCausalLM.from_config(cfg16, dtype=dtype)
                model.to_empty(device=device)
                missing, unexpected = _ckpt.load_prequantized_(model, str(model_name), device, dtype)
                if missing:
                    raise RuntimeError(f"{model_name}: checkpoint has no tensors for {missing[:8]} ...")
                model.config.quantization_config = _ckpt.bnb_quantization_config(dtype)
                prequantized = True
            else:
                model = AutoModelForCausalLM.from_pretrained(model_name, config=config, dtype=dtype)
                model.to(device)
            try:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(model_name)
            except Exception:
                tokenizer = None
        else:
            torch.manual_seed(random_state)
            old = torch.get_default_dtype()
            torch.set_default_dtype(dtype)
            try:
                with torch.device(device):
                    model = AutoModelForCausalLM.from_config(config)
            finally:
                torch.set_default_dtype(old)
            model.to(dtype)
        for p in model.parameters():
            p.requires_grad_(False)
        if load_in_4bit and not load_in_16bit and not prequantized:
            quantize_model_nf4_(model)
            torch.cuda.empty_cache()
        model.config.dtype = dtype
        FastLlamaModel.post_load(model, max_seq_length, dtype)
        from ..kernels import post_patch_loss_function
        post_patch_loss_function(model)
        FastLlamaModel.for_training(model, use_gradient_checkpointing)
        return model, tokenizer

    @staticmethod
    def get_peft_model(model, *args, **kwargs):
        return FastLlamaModel.get_peft_model(model, *args, **kwargs)

    @staticmethod
    def patch_peft_model(model, use_gradient_checkpointing="unsloth"):
        return FastLlamaModel.patch_peft_model(model, use_gradient_checkpointing)

    @staticmethod
    def for_training(model, use_gradient_checkpointing=True):
        return FastLlamaModel.for_training(model, use_gradient_checkpointing)

    @staticmethod
    def for_inference(model):
        return FastLlamaModel.for_inference(model)


class FastModel(FastLanguageModel):
    """loader.py:1140-2184 is the generic / VLM loader backed by unsloth_zoo's torch.compile path; here
    it is an alias of the language-model loader for the supported architectures."""
