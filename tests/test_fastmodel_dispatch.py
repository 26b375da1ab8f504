"""FastModel (models/loader.py): architecture dispatch, the VLM language-tower config mapping, and the tower checkpoint
reader -- CPU only."""
import os

import pytest
import torch


def _tiny_vl_config():
    from transformers import Qwen2VLConfig
    return Qwen2VLConfig(text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                                          num_key_value_heads=1, vocab_size=320, max_position_embeddings=256, rms_norm_eps=1e-6,
                                          rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
                                          tie_word_embeddings=False),
                         vision_config=dict(depth=1, embed_dim=32, hidden_size=256, num_heads=2))


def test_text_tower_config_of_qwen2_vl_keeps_shapes_and_mrope():
    from unsloth_amd.models.loader import text_tower_config
    from unsloth_amd.models.llama import _mrope_section
    vl = _tiny_vl_config()
    tc = text_tower_config(vl)
    assert tc.model_type == "qwen2"
    for k in ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads",
              "vocab_size", "rms_norm_eps"):
        assert getattr(tc, k) == getattr(vl.text_config, k), k
    assert _mrope_section(tc) == (16, 24, 24)
    assert tc.rope_parameters["rope_theta"] == 1e6
    assert text_tower_config(vl.text_config).hidden_size == 256          # the text config itself is accepted too
    from transformers import LlamaConfig, GPT2Config
    assert text_tower_config(LlamaConfig()) is None and text_tower_config(GPT2Config()) is None


def test_fastmodel_rejects_unsupported_architectures_loudly():
    from transformers import GPT2Config
    from unsloth_amd import FastModel
    with pytest.raises(NotImplementedError, match="gpt2"):
        FastModel.from_pretrained(config=GPT2Config())


def test_language_tower_checkpoint_reader(tmp_path):
    """A VLM checkpoint's `model.language_model.*` tensors land on the tower; vision tensors are ignored; missing ones reported."""
    from safetensors.torch import save_file
    from transformers import Qwen2ForCausalLM
    from unsloth_amd.checkpoint import load_language_tower_
    from unsloth_amd.models.loader import text_tower_config
    tc = text_tower_config(_tiny_vl_config())
    torch.manual_seed(0)
    src = Qwen2ForCausalLM(tc)
    sd = {}
    for k, v in src.state_dict().items():
        sd[("model.language_model." + k[len("model."):]) if k.startswith("model.") else k] = v.clone()
    sd["model.visual.patch_embed.proj.weight"] = torch.zeros(4, 4)
    dropped = "model.language_model.layers.1.mlp.up_proj.weight"
    kept_back = sd.pop(dropped)
    save_file(sd, os.path.join(tmp_path, "model.safetensors"))
    torch.manual_seed(1)
    dst = Qwen2ForCausalLM(tc)
    missing = load_language_tower_(dst, str(tmp_path))
    assert missing == ["model.layers.1.mlp.up_proj.weight"]
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        if k != "model.layers.1.mlp.up_proj.weight":
            assert torch.equal(a, b), k
    assert not torch.equal(dst.model.layers[1].mlp.up_proj.weight, kept_back)
