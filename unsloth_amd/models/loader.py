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
        if load_in_8bit:
            raise NotImplementedError(
                "8-bit loading goes through FastModel + the unsloth_zoo compiler in the reference "
                "(loader.py:487-523); outside the QLoRA hot path (SURVEY 8(f4)).")
        if full_finetuning and (load_in_4bit or load_in_16bit):
            # vision.py:1162-1168: "You selected full finetuning support, but 4bit / 8bit is enabled - disabling LoRA / QLoRA."
            load_in_4bit = load_in_16bit = False
        os.environ["UNSLOTH_ENABLE_FULL_FINETUNING"] = "1" if full_finetuning else "0"        # vision.py:1247-1268
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
            p.requires_grad_(bool(full_finetuning))        # vision.py:2206-2209: layernorms, embeddings and lm_head train too
        if full_finetuning and prequantized:
            raise ValueError("full_finetuning=True needs 16-bit weights; this checkpoint is pre-quantised to NF4")
        if load_in_4bit and not load_in_16bit and not prequantized:
            quantize_model_nf4_(model)
            torch.cuda.empty_cache()
        model.config.dtype = dtype
        FastLlamaModel.post_load(model, max_seq_length, dtype)
        from ..kernels import post_patch_loss_function
        post_patch_loss_function(model)
        FastLlamaModel.for_training(model, use_gradient_checkpointing)
        model._unsloth_full_finetuning = bool(full_finetuning)                                  # _utils.py:2899-2908
        if full_finetuning:
            FastLlamaModel.patch_full_finetune(model)
        return model, tokenizer

    @staticmethod
    def get_peft_model(model, *args, **kwargs):
        if getattr(model, "_unsloth_full_finetuning", False):
            print("Unsloth: Full finetuning is enabled, so .get_peft_model has no effect")      # vision.py:1884-1889
            return model
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


# VLM configs whose language tower is a Llama-family decoder the hand-kernel path covers (text_config.model_type -> the
# causal-LM architecture it is rebuilt as). Qwen2-VL / Qwen2.5-VL: a Qwen2 decoder (q/k/v bias, SwiGLU, RMSNorm) with
# multimodal RoPE -- three position streams over sections of the rotary dimension (BASELINE config 4).
VLM_TEXT_TOWERS = {"qwen2_vl_text": "qwen2", "qwen2_5_vl_text": "qwen2"}


def text_tower_config(config):
    """The causal-LM config of a supported VLM's language tower (mrope parameters kept), or None. `config` may be the
    VLM config (with `.text_config`) or the text config itself."""
    tc = getattr(config, "text_config", None) or config
    arch = VLM_TEXT_TOWERS.get(getattr(tc, "model_type", None))
    if arch is None:
        return None
    from transformers import Qwen2Config
    d = tc.to_dict()
    keep = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
            "num_key_value_heads", "hidden_act", "max_position_embeddings", "initializer_range", "rms_norm_eps",
            "use_cache", "tie_word_embeddings", "use_sliding_window", "sliding_window", "max_window_layers",
            "layer_types", "attention_dropout", "pad_token_id", "bos_token_id", "eos_token_id")
    kw = {k: d[k] for k in keep if k in d and d[k] is not None}
    rope = dict(d.get("rope_parameters") or {})
    legacy = d.get("rope_scaling") or {}
    if "mrope_section" not in rope and legacy.get("mrope_section"):
        rope["mrope_section"] = legacy["mrope_section"]
    if "mrope_section" not in rope:                      # Qwen2-VL's published split of the 64 rotary pairs
        half = (d["hidden_size"] // d["num_attention_heads"]) // 2
        rope["mrope_section"] = [half // 4, (half - half // 4) // 2, half - half // 4 - (half - half // 4) // 2]
    rope.setdefault("rope_type", "default")
    out = Qwen2Config(**kw, rope_parameters=rope)
    if not getattr(out, "use_sliding_window", False):
        out.sliding_window = None
    return out


class FastModel(FastLanguageModel):
    """loader.py:1140-2184: the reference's generic loader (text models of any architecture, VLMs, full fine-tuning),
    backed there by unsloth_zoo's torch.compile path. Here:
      * a language model of a supported architecture (llama / mistral / qwen2) -> the hand-kernel path (FastLanguageModel);
      * a VLM whose language tower is one (Qwen2-VL, Qwen2.5-VL) -> that TOWER as a causal LM on the hand-kernel path,
        multimodal RoPE included: position_ids [3, B, T] (temporal, height, width) with the config's `mrope_section`;
        text-only [B, T] positions work unchanged. The vision encoder is not built (its patch embeddings enter the
        tower as `inputs_embeds`, which the fast forward accepts); a local checkpoint directory is read with the
        `model.language_model.` prefix mapped onto the tower;
      * anything else raises, naming what the reference would do."""

    @staticmethod
    def from_pretrained(model_name=None, max_seq_length=2048, dtype=None, load_in_4bit=True, config=None,
                        full_finetuning=False, auto_model=None, *args, **kwargs):
        from transformers import AutoConfig
        if config is None and model_name is not None and os.path.isdir(str(model_name)):
            config = AutoConfig.from_pretrained(model_name, trust_remote_code=kwargs.get("trust_remote_code", False))
        if config is None:
            return FastLanguageModel.from_pretrained(model_name, max_seq_length, dtype, load_in_4bit, *args, **kwargs)
        if getattr(config, "model_type", None) in SUPPORTED:
            return FastLanguageModel.from_pretrained(model_name, max_seq_length, dtype, load_in_4bit, config=config,
                                                     full_finetuning=full_finetuning, *args, **kwargs)
        tower = text_tower_config(config)
        if tower is None:
            raise NotImplementedError(
                f"FastModel: model_type={getattr(config, 'model_type', None)!r} has no hand-kernel path here "
                f"(supported: {SUPPORTED} and the language towers of {sorted(VLM_TEXT_TOWERS)}). The reference sends "
                "other architectures through unsloth_zoo's compiled generic path (loader.py:1140-2184), which is out "
                "of scope (SURVEY 8(f4)).")
        if model_name is not None and os.path.isdir(str(model_name)):
            # weights of the tower only: build from the mapped config, then fill from the VLM checkpoint's language_model keys
            model, tok = FastLanguageModel.from_pretrained(None, max_seq_length, dtype, False, config=tower,
                                                           full_finetuning=full_finetuning, *args, **kwargs)
            from .. import checkpoint as _ckpt
            missing = _ckpt.load_language_tower_(model, str(model_name))
            if missing:
                raise RuntimeError(f"{model_name}: no tensors for {missing[:8]} ... in the checkpoint's language tower")
            if load_in_4bit:
                quantize_model_nf4_(model)
                FastLlamaModel.post_load(model, max_seq_length, model.config.dtype)
            return model, tok
        return FastLanguageModel.from_pretrained(None, max_seq_length, dtype, load_in_4bit, config=tower,
                                                 full_finetuning=full_finetuning, *args, **kwargs)


from .vision import FastVisionModel    # noqa: E402  (loader.py:2187 aliases FastVisionModel = FastModel; here the VLM wrapper)
