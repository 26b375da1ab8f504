"""CPU: the class-level patch points (unsloth/models/llama.py:2300-2319) and `_patch_trl_trainer`
(unsloth/trainer.py:988-1021).
  * after FastLlamaModel.pre_patch() the HF classes carry our adapters, stock models (not prepared by the loader)
    still compute exactly what they computed before (the adapters fall through), and unpatch_all() restores them;
  * `_patch_trl_trainer` moves pre-0.13 keyword arguments onto the config and renames `tokenizer`, exercised on a
    stand-in `trl` module (TRL itself is not installed in this image)."""
import dataclasses
import sys
import types

import pytest
import torch


def _tiny():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=16, vocab_size=100, max_position_embeddings=64)
    return LlamaForCausalLM(cfg).eval()


def test_class_level_patches_fall_through_for_stock_models_and_unpatch():
    from transformers.models.llama import modeling_llama as m
    from unsloth_amd.kernels import unpatch_rms_layernorm
    from unsloth_amd.models import llama as L
    model = _tiny()
    ids = torch.randint(0, 100, (2, 10))
    with torch.no_grad():
        before = model(input_ids=ids)
    orig = (m.LlamaAttention.forward, m.LlamaDecoderLayer.forward, m.LlamaModel.forward, m.LlamaForCausalLM.forward)
    L.FastLlamaModel.pre_patch()
    try:
        now = (m.LlamaAttention.forward, m.LlamaDecoderLayer.forward, m.LlamaModel.forward, m.LlamaForCausalLM.forward)
        assert all(a is not b for a, b in zip(orig, now)), "a class-level forward was not replaced"
        assert (m.LlamaAttention, "forward") in L._PATCHED and (m.LlamaDecoderLayer, "forward") in L._PATCHED
        with torch.no_grad():
            after = model(input_ids=ids)
            # a layer called directly, the way code that bypasses the CausalLM does
            h = model.model.embed_tokens(ids)
            pos = model.model.rotary_emb(h, torch.arange(10)[None])
            a = model.model.layers[0](h, position_embeddings=pos)
        assert torch.equal(before.logits, after.logits)
        L.unpatch_all()
        with torch.no_grad():
            b = model.model.layers[0](h, position_embeddings=pos)
        assert torch.equal(a if torch.is_tensor(a) else a[0], b if torch.is_tensor(b) else b[0])
        assert (m.LlamaAttention.forward, m.LlamaDecoderLayer.forward, m.LlamaModel.forward,
                m.LlamaForCausalLM.forward) == orig
    finally:
        from unsloth_amd.kernels.cross_entropy_loss import unpatch_loss_functions
        L.unpatch_all()
        unpatch_rms_layernorm()
        unpatch_loss_functions()


def test_patch_trl_trainer_on_a_stand_in_module(monkeypatch):
    from transformers import TrainingArguments
    from unsloth_amd import trainer as T

    @dataclasses.dataclass
    class SFTConfig(TrainingArguments):
        max_seq_length: int = 1024
        dataset_text_field: str = "text"
        packing: bool = False

    class SFTTrainer:
        def __init__(self, model=None, args=None, train_dataset=None, processing_class=None):
            self.model, self.args, self.processing_class = model, args, processing_class

    class DPOTrainer:                      # no DPOConfig in the stand-in: must be left alone
        def __init__(self, model=None):
            self.model = model

    trl = types.ModuleType("trl")
    trl.__version__ = "0.19.0"
    trl.trainer = types.ModuleType("trl.trainer")
    for mod in (trl, trl.trainer):
        mod.SFTConfig, mod.SFTTrainer, mod.DPOTrainer = SFTConfig, SFTTrainer, DPOTrainer
    monkeypatch.setitem(sys.modules, "trl", trl)
    monkeypatch.setitem(sys.modules, "trl.trainer", trl.trainer)
    dpo_init = DPOTrainer.__init__
    assert T._patch_trl_trainer() == ["SFT"]
    assert getattr(trl, "__UNSLOTH_BACKWARDS_COMPATIBLE__") is True and DPOTrainer.__init__ is dpo_init
    assert T._patch_trl_trainer() == []            # idempotent
    cfg = SFTConfig(output_dir="/tmp/x", report_to=[])
    t = SFTTrainer(model="m", args=cfg, tokenizer="tok", max_seq_length=2048, dataset_text_field="body", packing=True)
    assert t.processing_class == "tok" and t.args is cfg
    assert (cfg.max_seq_length, cfg.dataset_text_field, cfg.packing) == (2048, "body", True)
    # the current calling convention passes through untouched
    t2 = SFTTrainer(model="m", args=SFTConfig(output_dir="/tmp/x", report_to=[]), processing_class="p")
    assert t2.processing_class == "p" and t2.args.max_seq_length == 1024


def test_patch_trl_trainer_without_trl_is_a_no_op():
    from unsloth_amd import trainer as T
    if "trl" in sys.modules and not hasattr(sys.modules["trl"], "SFTTrainer"):
        del sys.modules["trl"]
    try:
        import trl  # noqa: F401
        return          # a real TRL is present: covered by the stand-in test's logic
    except Exception:
        assert T._patch_trl_trainer() == []


def test_recompute_policy_schedules():
    """"all*3,attn": a per-layer schedule between two selective-recompute policies (models/fast_layer.py)."""
    from unsloth_amd.models.fast_layer import POLICIES, policy_for_layer, resolve_policy_spec
    assert resolve_policy_spec("attn") == POLICIES["attn"]
    assert resolve_policy_spec("qkv+eg") == frozenset({"qkv", "eg"})
    sched = resolve_policy_spec("all*3,min*2,attn")
    got = [policy_for_layer(sched, i) for i in range(8)]
    assert got[:3] == [POLICIES["all"]] * 3 and got[3:5] == [POLICIES["min"]] * 2 and got[5:] == [POLICIES["attn"]] * 3
    assert policy_for_layer(POLICIES["min"], 7) == POLICIES["min"]


# ---- the PEFT plugin contract of get_lora_parameters(_bias): the reference's tests/test_fast_gemv_dispatch.py:38-63 ----
def _proj(weight, weight_scale=None, **extra):
    from types import SimpleNamespace
    proj = SimpleNamespace(weight=weight, bias=None, merged=False, **extra)
    if weight_scale is not None:
        proj.weight_scale = weight_scale
    return proj


def test_bf16_weight_scale_is_not_a_quant_state():
    """A bf16 weight that still carries a weight_scale (a decompressed compressed-tensors layer): quant state None, for
    both accessors."""
    from unsloth_amd.kernels.utils import get_lora_parameters, get_lora_parameters_bias
    proj = _proj(torch.randn(4, 4, dtype=torch.bfloat16), torch.rand(2, 2))
    assert get_lora_parameters_bias(proj)[1] is None
    assert get_lora_parameters(proj)[1] is None
    assert get_lora_parameters_bias(_proj(torch.randn(4, 4, dtype=torch.bfloat16)))[1] is None


def test_fp8_weight_keeps_its_scale_and_the_kernels_refuse_it():
    from unsloth_amd.kernels.utils import (get_lora_parameters, get_lora_parameters_bias, _FP8_WEIGHT_DTYPES, matmul_lora,
                                           fast_dequantize)
    if not _FP8_WEIGHT_DTYPES:
        pytest.skip("no float8 dtype in this torch build")
    scale = torch.rand(2, 2)
    proj = _proj(torch.randn(4, 4).to(_FP8_WEIGHT_DTYPES[0]), scale)
    W, q = get_lora_parameters_bias(proj)[:2]
    assert q is scale and get_lora_parameters(proj)[1] is scale
    # weight_scale_inv wins over weight_scale; quant_method "fp8" stamps the block size on weight and state (utils.py:351-366)
    inv = torch.rand(2, 2)
    proj2 = _proj(torch.randn(4, 4).to(_FP8_WEIGHT_DTYPES[0]), scale, weight_scale_inv=inv, quant_method="fp8")
    W2, q2 = get_lora_parameters(proj2)[:2]
    assert q2 is inv and W2.block_size == [128, 128] and q2.block_size == [128, 128]
    # no fp8 GEMM on this backend: loud refusal instead of misreading the scale as an NF4 quant state
    with pytest.raises(NotImplementedError, match="fp8"):
        fast_dequantize(W, q)
    with pytest.raises(NotImplementedError, match="fp8"):
        matmul_lora(torch.randn(1, 3, 4, dtype=torch.bfloat16), W, q, None, None, None)
