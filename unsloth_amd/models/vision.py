"""FastVisionModel for Qwen2-VL / Qwen2.5-VL (BASELINE config 4).

Reference: unsloth/models/vision.py:881-1990 -- `FastBaseModel.from_pretrained` loads the HF VLM and hands it to
unsloth_zoo's compiler; `get_peft_model` (:1855-1990) places LoRA by regex (`finetune_vision_layers`,
`finetune_language_layers`, `finetune_attention_modules`, `finetune_mlp_modules`). Neither the compiler nor the regex
helper is in the repository (third party): the behaviour restated here is "HF's Qwen2-VL composition, with the language
tower on the fused path".

MI355X composition:
  * language tower = the Qwen2 causal LM on the hand-kernel path (models/loader.py FastModel: NF4, LoRA through the
    grouped MFMA GEMMs with the q/k/v bias in the epilogue, multimodal RoPE kernel, flash attention, fused linear-CE);
  * vision tower = transformers' Qwen2VisionTransformerPretrainedModel's modules and parameters (patch-embed, ViT blocks,
    patch merger) with the blocks' forwards replaced (models/vision_tower.py, round 4): 2-D RoPE through
    csrc/rope_embedding.hip, NON-CAUSAL attention inside the `cu_seqlens` windows through csrc/attention.hip, QuickGELU
    through csrc/glu.hip; the frozen LayerNorms run through `fast_layernorm` (csrc/layernorm.hip) and, once LoRA is
    attached, the linear layers (qkv / proj / fc1 / fc2) through LoRA_W, i.e. the MFMA GEMM kernels with bias in the epilogue;
  * glue (this file): image features scattered over the image placeholder tokens (`masked_scatter`, as HF), and the
    [3, B, T] multimodal position ids -- INTEGER work, restated from transformers' `get_rope_index` /
    `get_vision_position_ids` and tested bit-exact against them (tests/test_vision.py).
"""
import itertools
import os

import torch
import torch.nn as nn
from transformers.modeling_outputs import CausalLMOutputWithPast

from .. import lora as _lora


# ------------------------------------------------------------------------------------------------------------------
def vision_position_ids(start, grid_thw, spatial_merge_size, device=None):
    """[3, t * (h/m) * (w/m)] positions of one image / video block that starts at text position `start`: temporal index,
    row, column of every merged patch, each offset by `start` (transformers `get_vision_position_ids` with
    temporal step 1, the Qwen2-VL rule)."""
    t, h, w = (int(x) for x in grid_thw)
    hh, ww = h // spatial_merge_size, w // spatial_merge_size
    ti = torch.arange(t, device=device).view(t, 1, 1).expand(t, hh, ww).reshape(-1)
    hi = torch.arange(hh, device=device).view(1, hh, 1).expand(t, hh, ww).reshape(-1)
    wi = torch.arange(ww, device=device).view(1, 1, ww).expand(t, hh, ww).reshape(-1)
    return torch.stack([ti, hi, wi]) + start


def mrope_position_ids(input_ids, image_grid_thw=None, video_grid_thw=None, attention_mask=None, image_token_id=None,
                       video_token_id=None, spatial_merge_size=2):
    """Multimodal RoPE positions [3, B, T] (temporal, height, width) + the per-row delta `max position + 1 - length`,
    restating transformers' Qwen2VLModel.get_rope_index: runs of text tokens count up by one in all three streams; an
    image (video) placeholder run takes the positions of its merged-patch grid; the text after it resumes at
    start + max(h, w) / merge. Padding (attention_mask == 0) is skipped and gets position 0. Integer work: exact."""
    B, T = input_ids.shape
    dev = input_ids.device
    pos = torch.zeros(3, B, T, dtype=input_ids.dtype, device=dev)
    grids = {1: iter(image_grid_thw) if image_grid_thw is not None else None,
             2: iter(video_grid_thw) if video_grid_thw is not None else None}
    deltas = []
    for b in range(B):
        ids = input_ids[b]
        keep = attention_mask[b].bool() if attention_mask is not None else None
        if keep is not None:
            ids = ids[keep]
        kinds = torch.zeros_like(ids)
        if image_token_id is not None:
            kinds = torch.where(ids == image_token_id, torch.ones_like(kinds), kinds)
        if video_token_id is not None:
            kinds = torch.where(ids == video_token_id, torch.full_like(kinds, 2), kinds)
        cur, parts, i = 0, [], 0
        for kind, grp in itertools.groupby(kinds.tolist()):
            n = len(list(grp))
            if kind == 0:
                parts.append(torch.arange(n, device=dev).view(1, -1).expand(3, -1) + cur)
                cur += n
            else:
                g = next(grids[kind])
                vp = vision_position_ids(cur, g, spatial_merge_size, dev)
                if vp.shape[1] != n:
                    raise ValueError(f"row {b}: {n} placeholder tokens for a grid of {vp.shape[1]} merged patches")
                parts.append(vp)
                cur += max(int(g[1]), int(g[2])) // spatial_merge_size
            i += n
        p = torch.cat(parts, dim=1).reshape(3, -1).to(pos.dtype)
        if keep is not None:
            pos[:, b, keep] = p
        else:
            pos[:, b] = p
        deltas.append(int(p.max()) + 1 - ids.shape[0])
    return pos, torch.tensor(deltas, device=dev).unsqueeze(1)


# ------------------------------------------------------------------------------------------------------------------
class Qwen2VLFastModel(nn.Module):
    """vision tower + language tower (fused path). LOADING takes HF's Qwen2VLForConditionalGeneration names (`model.visual.*`
    through checkpoint.load_prefixed_, the language tower through the text loader). state_dict() of THIS module emits its own
    tree -- `visual.*`, `language.model.*` / `language.lm_head.*` (`language.base_model.model.*` once LoRA is attached) -- and
    is not an HF checkpoint: save adapters with the language tower's PEFT model and the ViT's LoRA factors by name; a merged HF
    export of the VLM is the reference's save pipeline, out of scope (GGUF / save paths)."""

    def __init__(self, config, visual, language):
        super().__init__()
        self.config = config
        self.visual = visual
        self.language = language                     # Qwen2ForCausalLM on the hand-kernel path
        self.rope_deltas = None

    # --- HF-style accessors ---------------------------------------------------------------------------------------
    @property
    def lm_head(self):
        return self.language.lm_head

    def get_input_embeddings(self):
        return self.language.model.embed_tokens

    def get_output_embeddings(self):
        return self.language.lm_head

    def get_image_features(self, pixel_values, image_grid_thw):
        """[sum of merged patches, hidden]: the patch merger's output for every image, in order."""
        pixel_values = pixel_values.to(self.visual.get_dtype())
        out = self.visual(pixel_values, grid_thw=image_grid_thw)
        return out.pooler_output if hasattr(out, "pooler_output") else out

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None, labels=None,
                pixel_values=None, pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None, **kwargs):
        cfg = self.config
        if inputs_embeds is None:
            inputs_embeds = self.get_input_embeddings()(input_ids)
        for pv, grid, tok in ((pixel_values, image_grid_thw, cfg.image_token_id),
                              (pixel_values_videos, video_grid_thw, cfg.video_token_id)):
            if pv is None:
                continue
            feats = self.get_image_features(pv, grid).to(inputs_embeds.device, inputs_embeds.dtype)
            mask = (input_ids == tok)
            if int(mask.sum()) != feats.shape[0]:
                raise ValueError(f"{int(mask.sum())} placeholder tokens but {feats.shape[0]} vision features")
            inputs_embeds = inputs_embeds.masked_scatter(mask.unsqueeze(-1).expand_as(inputs_embeds), feats)
        if position_ids is None and input_ids is not None:
            position_ids, self.rope_deltas = mrope_position_ids(
                input_ids, image_grid_thw, video_grid_thw, attention_mask, cfg.image_token_id, cfg.video_token_id,
                cfg.vision_config.spatial_merge_size)
        return self.language(input_ids=None, inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                             position_ids=position_ids, labels=labels, **kwargs)

    # PEFT-style helpers the trainer surface uses
    def get_base_model(self):
        return self

    def for_training(self, use_gradient_checkpointing=True):
        from .llama import FastLlamaModel
        FastLlamaModel.for_training(self.language, use_gradient_checkpointing)
        self.train()
        return self

    def for_inference(self):
        self.eval()
        return self


def _vision_targets(finetune_attention_modules, finetune_mlp_modules):
    t = []
    if finetune_attention_modules:
        t += ["qkv", "proj"]
    if finetune_mlp_modules:
        t += ["fc1", "fc2"]
    return t


class FastVisionModel:
    """`FastVisionModel.from_pretrained(...)` / `.get_peft_model(...)` (unsloth/models/vision.py:881, :1855)."""

    @staticmethod
    def from_pretrained(model_name=None, max_seq_length=2048, dtype=None, load_in_4bit=True, config=None,
                        full_finetuning=False, device=None, random_state=3407, use_gradient_checkpointing="unsloth",
                        **kwargs):
        from .loader import FastModel, text_tower_config, _resolve_dtype
        from ..kernels.layernorm import patch_layernorm
        from transformers import AutoConfig
        from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VisionTransformerPretrainedModel
        if config is None and model_name is not None and os.path.isdir(str(model_name)):
            config = AutoConfig.from_pretrained(model_name)
        if config is None or text_tower_config(config) is None or not hasattr(config, "vision_config"):
            raise NotImplementedError("FastVisionModel: Qwen2-VL / Qwen2.5-VL style configs (a supported language tower + "
                                      "`vision_config`); other VLMs take the reference's compiled generic path, out of scope")
        dt = _resolve_dtype(dtype)
        language, tok = FastModel.from_pretrained(model_name, max_seq_length, dtype, load_in_4bit, config=config,
                                                  full_finetuning=full_finetuning, device=device,
                                                  random_state=random_state,
                                                  use_gradient_checkpointing=use_gradient_checkpointing, **kwargs)
        dev = next(language.parameters()).device
        torch.manual_seed(random_state + 1)
        patch_layernorm()                           # frozen LayerNorms of the ViT -> csrc/layernorm.hip
        vcfg = config.vision_config
        if getattr(vcfg, "model_type", "") not in ("qwen2_vl", "qwen2_vl_vision"):
            # Qwen2.5-VL's tower (window attention, RMSNorm, SwiGLU) is a different module tree
            raise NotImplementedError(f"vision tower {getattr(vcfg, 'model_type', None)!r}: only Qwen2-VL's ViT is built here")
        with torch.device(dev):
            visual = Qwen2VisionTransformerPretrainedModel._from_config(vcfg) if hasattr(
                Qwen2VisionTransformerPretrainedModel, "_from_config") else Qwen2VisionTransformerPretrainedModel(vcfg)
        visual.to(dt)
        if model_name is not None and os.path.isdir(str(model_name)):
            from .. import checkpoint as _ckpt
            missing = _ckpt.load_prefixed_(visual, str(model_name), ("model.visual.", "visual."))
            if missing:
                raise RuntimeError(f"{model_name}: no tensors for {missing[:8]} ... in the checkpoint's vision tower")
        for p in visual.parameters():
            p.requires_grad_(False)
        from .vision_tower import patch_vision_tower
        patch_vision_tower(visual)                  # ViT attention (2-D RoPE + non-causal flash) and QuickGELU on the HIP kernels
        model = Qwen2VLFastModel(config, visual, language)
        model.max_seq_length = max_seq_length
        from ._utils import prepare_for_trainer
        prepare_for_trainer(model)                  # vision.py:1797-1823, :2162 -- the marker and the DDP ignore list on the wrapper Trainer sees
        return model, tok

    @staticmethod
    def get_peft_model(model, r=16, target_modules=None, lora_alpha=16, lora_dropout=0.0, bias="none",
                       finetune_vision_layers=True, finetune_language_layers=True, finetune_attention_modules=True,
                       finetune_mlp_modules=True, use_gradient_checkpointing="unsloth", random_state=3407,
                       use_rslora=False, init_lora_weights=True, **kwargs):
        """vision.py:1855-1990. The language tower gets the fused LoRA path (FastLlamaModel.get_peft_model on the q/k/v/o
        and / or gate/up/down projections); the vision tower's linears are wrapped in LoraLayer (forward through LoRA_W)."""
        from .llama import FastLlamaModel
        if type(r) is not int:
            raise TypeError(f"Unsloth: Rank of {str(r)} must be an integer.")
        if r <= 0:
            raise TypeError(f"Unsloth: Rank of {str(r)} must be larger than 0.")
        if not isinstance(model, Qwen2VLFastModel):
            raise TypeError("FastVisionModel.get_peft_model expects the model FastVisionModel.from_pretrained returned")
        if isinstance(model.language, _lora.PeftModelForCausalLM):
            raise RuntimeError("Unsloth: You already added LoRA adapters to your model!")
        if target_modules == "all-linear":
            finetune_vision_layers = finetune_language_layers = finetune_attention_modules = finetune_mlp_modules = True
        if finetune_language_layers:
            lang_targets = ([] if not finetune_attention_modules else ["q_proj", "k_proj", "v_proj", "o_proj"]) + \
                           ([] if not finetune_mlp_modules else ["gate_proj", "up_proj", "down_proj"])
            if isinstance(target_modules, (list, tuple)):
                lang_targets = [t for t in target_modules if t in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj",
                                                                   "up_proj", "down_proj")] or lang_targets
            model.language = FastLlamaModel.get_peft_model(
                model.language, r=r, target_modules=lang_targets, lora_alpha=lora_alpha, lora_dropout=lora_dropout,
                bias=bias, use_gradient_checkpointing=use_gradient_checkpointing, random_state=random_state,
                use_rslora=use_rslora, init_lora_weights=init_lora_weights)
        else:
            FastLlamaModel.for_training(model.language, use_gradient_checkpointing)
        if finetune_vision_layers:
            torch.manual_seed(random_state + 2)
            targets = _vision_targets(finetune_attention_modules, finetune_mlp_modules)
            for name, module in list(model.visual.named_modules()):
                leaf = name.rsplit(".", 1)[-1]
                if leaf not in targets or not isinstance(module, nn.Linear) or ".blocks." not in "." + name:
                    continue
                parent = model.visual.get_submodule(name.rsplit(".", 1)[0])
                setattr(parent, leaf, _lora.LoraLayer(module, "default", r, lora_alpha, lora_dropout, use_rslora,
                                                      init_lora_weights))
        model.train()
        from ._utils import prepare_for_trainer
        return prepare_for_trainer(model)           # vision.py:2257: the buffers' names changed under the PEFT wrapper
