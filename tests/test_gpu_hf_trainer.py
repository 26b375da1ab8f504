"""-m gpu: the patched model under HuggingFace's STOCK `transformers.Trainer` -- the reference's real hot loop
(trainer.py:502-623 -> Trainer.training_step -> compute_loss -> PeftModel_fast_forward, models/_utils.py:3142-3313):
FastLanguageModel.from_pretrained -> get_peft_model -> Trainer(bf16 autocast, gradient accumulation 2, gradient_checkpointing
=True, Trainer's own torch AdamW) on a pre-tokenised padding-free dataset, every kernel on the HIP path. The very batches the
Trainer consumed are then replayed through unsloth_amd.trainer's own pieces (FlatAdamW, hand-written accumulation) on a twin
model: the per-step losses agree within 1e-3 (relative), and `num_items_in_batch` was the whole window's shifted count."""
import pytest
import torch

from tests._hf_trainer_util import Docs, PaddedCollator, PaddingFreeCollator, replay, run_stock_trainer, shifted_targets

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _tiny(gc, load_in_4bit=True, seed=3407):
    from transformers import LlamaConfig
    from unsloth_amd import FastLanguageModel
    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=128, vocab_size=1000, rms_norm_eps=1e-5,
                      max_position_embeddings=512, rope_parameters={"rope_type": "default", "rope_theta": 5e5},
                      tie_word_embeddings=False)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=512, load_in_4bit=load_in_4bit,
                                                 device=DEV, random_state=seed, use_gradient_checkpointing=gc)
    model = FastLanguageModel.get_peft_model(model, r=8, lora_alpha=16, use_gradient_checkpointing=gc, random_state=seed)
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.05).to(DEV))
    return model


@pytest.mark.parametrize("gc,collator_kind", [("unsloth", "padding_free"), (True, "padding_free"), ("unsloth", "padded")])
def test_stock_trainer_matches_unsloth_train_pieces(tmp_path, gc, collator_kind):
    from unsloth_amd.kernels import utils as U
    from unsloth_amd.trainer import make_optimizer
    model = _tiny(gc)
    assert model.get_base_model()._unsloth_amd_patched == (2, 2, 2)
    data = Docs(64, 1000, 20, 90)
    collator = PaddingFreeCollator() if collator_kind == "padding_free" else PaddedCollator(96)
    seen_items = []
    inner = model.get_base_model()
    hook = inner.register_forward_pre_hook(lambda m, a, kw: seen_items.append(kw.get("num_items_in_batch")), with_kwargs=True)
    launches = []
    orig_launch = U._launch_gemm

    def counting(*a, **kw):
        launches.append(1)
        return orig_launch(*a, **kw)
    U._launch_gemm = counting
    try:
        losses, trainer = run_stock_trainer(model, data, collator, tmp_path, bf16=True)
    finally:
        U._launch_gemm = orig_launch
        hook.remove()
    assert len(launches) > 0, "the Trainer's forward never reached the HIP GEMM"
    assert len(losses) == 3 and trainer.state.global_step == 3 and trainer.model_accepts_loss_kwargs
    assert inner.model.gradient_checkpointing                                     # TrainingArguments(gradient_checkpointing=True)
    used = collator.seen[:6]
    assert len(seen_items) == 6
    for step in range(3):
        want = sum(shifted_targets(b) for b in used[2 * step:2 * step + 2])
        assert all(int(n) == want for n in seen_items[2 * step:2 * step + 2]), (step, seen_items, want)
    twin = _tiny(gc)
    want = replay(twin, used, DEV, optimizer=make_optimizer(twin, lr=2e-4))
    for a, b in zip(losses, want):
        assert abs(a - b) <= 1e-3 * abs(b) + 1e-4, (losses, want)                # +1e-4: the Trainer logs 4 decimals
    assert all(torch.isfinite(torch.tensor(losses)))
    # both trained the same way: the adapters end up close (bf16 step-to-step noise only)
    a = torch.cat([p.detach().flatten() for n, p in sorted(model.named_parameters()) if p.requires_grad])
    b = torch.cat([p.detach().flatten() for n, p in sorted(twin.named_parameters()) if p.requires_grad])
    assert float((a - b).norm() / b.norm()) < 2e-3


def test_evaluate_and_dense_16bit_model_under_the_stock_trainer(tmp_path):
    """load_in_4bit=False (16-bit LoRA) through the same loop, then Trainer.evaluate(): an eval pass under no_grad returns a loss."""
    model = _tiny("unsloth", load_in_4bit=False)
    data = Docs(32, 1000, 20, 90, seed=5)
    collator = PaddingFreeCollator()
    losses, trainer = run_stock_trainer(model, data, collator, tmp_path, bf16=True, steps=2)
    assert len(losses) == 2
    trainer.args.prediction_loss_only = True          # the fused-CE path hands back EMPTY_LOGITS, as the reference's does
    metrics = trainer.evaluate(eval_dataset=Docs(6, 1000, 20, 90, seed=9))
    assert "eval_loss" in metrics and 0.0 < metrics["eval_loss"] < 20.0
