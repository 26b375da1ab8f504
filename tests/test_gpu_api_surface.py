"""-m gpu: corners of the drop-in surface a user of the reference reaches without thinking about them -- LoRA dropout,
fp16 models, gradient accumulation (plain autograd and the flat AdamW arena), evaluation under no_grad -- each against the
fp32 HF oracle or against the same computation done the long way."""
import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _cfg(head_dim=128):
    from transformers import LlamaConfig
    return LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4 if head_dim == 128 else 8,
                       num_key_value_heads=2, head_dim=head_dim, vocab_size=1000, rms_norm_eps=1e-5, max_position_embeddings=512,
                       rope_parameters={"rope_type": "default", "rope_theta": 5e5}, tie_word_embeddings=False)


def _model(dtype=None, dropout=0.0, gc=False, load_in_4bit=True):
    from unsloth_amd import FastLanguageModel
    model, _ = FastLanguageModel.from_pretrained(config=_cfg(), max_seq_length=256, dtype=dtype, load_in_4bit=load_in_4bit,
                                                 device=DEV, random_state=3407, use_gradient_checkpointing=gc)
    model = FastLanguageModel.get_peft_model(model, r=8, lora_alpha=16, lora_dropout=dropout, use_gradient_checkpointing=gc,
                                             random_state=3407)
    g = torch.Generator().manual_seed(21)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.05).to(DEV))
    return model


def _batch(seed, B=2, T=96):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 1000, (B, T), generator=g)
    labels = ids.clone()
    labels[0, :7] = -100
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0).expand(B, T).contiguous()
    return ids, labels, pos


def _grads(model):
    return {"layers." + n.split(".layers.", 1)[1].replace(".default.weight", ""): p.grad.detach().float().cpu()
            for n, p in model.named_parameters() if p.requires_grad}


def _errors(got, ref):
    worst = max((rel_fro(got[k], ref[k]), k) for k in got)
    total = rel_fro(torch.cat([got[k].flatten() for k in sorted(got)]), torch.cat([ref[k].flatten() for k in sorted(got)]))
    return worst, total


def test_fp16_model_matches_the_oracle():
    """dtype=torch.float16 (the reference's default on GPUs without bf16): every kernel has an fp16 instantiation; fp16 has
    three more mantissa bits than bf16, so the bounds are tighter than the bf16 ones."""
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    model = _model(dtype=torch.float16)
    assert next(p for n, p in model.named_parameters() if "norm" in n).dtype == torch.float16
    ids, labels, pos = _batch(1)
    out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
    out.loss.backward()
    ref_loss, ref = hf_reference_loss_and_lora_grads(model, ids, labels, pos)
    assert abs(float(out.loss) - float(ref_loss)) <= 5e-4 * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    (worst, wk), total = _errors(_grads(model), ref)
    assert worst < 1.5e-2 and total < 3e-3, (worst, wk, total)          # measured 8.5e-3 (one small q_proj.lora_A) / 1.0e-3


def test_lora_dropout_takes_the_unfused_path_and_trains():
    """lora_dropout > 0: the reference installs no fused hook (llama.py:3695-3772) and PEFT's forward runs; here LoraLayer's own
    forward (base through the MFMA GEMM Function, the adapter through torch with dropout). In eval mode dropout is off: the
    oracle's numbers; in train mode: finite non-zero gradients everywhere, different masks -> different losses."""
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    model = _model(dropout=0.25)
    assert model.get_base_model()._unsloth_amd_patched == (0, 0, 0)
    ids, labels, pos = _batch(2)
    model.eval()
    with torch.enable_grad():
        out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
        out.loss.backward()
    ref_loss, ref = hf_reference_loss_and_lora_grads(model, ids, labels, pos)
    assert abs(float(out.loss) - float(ref_loss)) <= 1e-3 * abs(float(ref_loss))
    (worst, wk), total = _errors(_grads(model), ref)
    assert worst < 2.5e-2 and total < 1.5e-2, (worst, wk, total)
    model.train()
    losses = []
    for _ in range(2):
        for p in model.parameters():
            p.grad = None
        out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
        out.loss.backward()
        losses.append(float(out.loss))
        gs = [p.grad for p in model.parameters() if p.requires_grad]
        assert all(g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0 for g in gs)
    assert losses[0] != losses[1] and abs(losses[0] - float(ref_loss)) < 0.2 * abs(float(ref_loss))


@pytest.mark.parametrize("flat", [False, True])
def test_gradient_accumulation_over_micro_batches(flat):
    """Two micro-batches, one optimizer step: the accumulated gradient is the sum of the two (autograd's AccumulateGrad, or
    -- with optim.FlatAdamW -- the fused LoRA-gradient kernel ADDING into the flat arena), each pre-normalised by the global
    token count like HF Trainer does (num_items_in_batch)."""
    from unsloth_amd.trainer import make_optimizer
    model = _model()
    b1, b2 = _batch(3), _batch(4)
    n_items = int((b1[1][:, 1:] != -100).sum() + (b2[1][:, 1:] != -100).sum())
    opt = make_optimizer(model, lr=1e-3, flat=flat)
    assert (getattr(opt, "flat_p", None) is not None) == flat

    def run(b):
        ids, labels, pos = b
        out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV), num_items_in_batch=n_items)
        out.loss.backward()
        return float(out.loss)
    opt.zero_grad()
    run(b1)
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    opt.zero_grad()
    run(b2)
    g2 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    opt.zero_grad()
    run(b1)
    run(b2)
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert rel_fro(p.grad.float(), (g1[n] + g2[n]).float()) < 1e-5, n
    before = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad]).clone()
    opt.step()
    opt.zero_grad()
    after = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad])
    assert torch.isfinite(after).all() and float((after - before).abs().max()) > 0
    if flat:
        assert float(opt.arena.arena.abs().max()) == 0.0         # the step zeroed the arena: the next accumulation starts clean


def test_evaluation_under_no_grad_equals_the_training_forward():
    model = _model(gc="unsloth")
    ids, labels, pos = _batch(5)
    kw = dict(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
    train_loss = float(model(**kw).loss)
    model.eval()
    with torch.no_grad():
        eval_loss = float(model(**kw).loss)
    model.train()
    assert abs(train_loss - eval_loss) <= 1e-6 * abs(train_loss)
