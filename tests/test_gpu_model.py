"""-m gpu: the whole drop-in surface (FastLanguageModel.from_pretrained -> get_peft_model -> forward/backward)
through the HIP path against the implementation-independent oracle (stock HF model on the CPU, fp32, merged
LoRA over oracle-dequantised NF4 weights)."""
import os

import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _tiny(load_in_4bit=True, gc=True, r=8, layers=2, seed=3407, head_dim=32):
    from transformers import LlamaConfig
    from unsloth_amd import FastLanguageModel
    heads = 8 if head_dim == 32 else 4            # head_dim 128: the hand-written flash-attention kernels run
    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=layers, num_attention_heads=heads,
                      num_key_value_heads=2, head_dim=head_dim, vocab_size=1000, rms_norm_eps=1e-5,
                      max_position_embeddings=512, rope_parameters={"rope_type": "default", "rope_theta": 5e5},
                      tie_word_embeddings=False)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=256, load_in_4bit=load_in_4bit,
                                                 device=DEV, random_state=seed, use_gradient_checkpointing=gc)
    model = FastLanguageModel.get_peft_model(model, r=r, lora_alpha=2 * r, use_gradient_checkpointing=gc,
                                             random_state=seed)
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.05).to(DEV))
    return model


def _batch(B=2, T=96, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 1000, (B, T), generator=g)
    labels = ids.clone()
    labels[0, :5] = -100
    pos = torch.arange(T, dtype=torch.int32).unsqueeze(0).expand(B, T).contiguous()
    return ids, labels, pos


def _grads(model):
    return {"layers." + n.split(".layers.", 1)[1].replace(".default.weight", ""): p.grad.detach().float().cpu()
            for n, p in model.named_parameters() if p.requires_grad}


@pytest.mark.parametrize("load_in_4bit,head_dim", [(True, 32), (False, 32), (True, 128)])
def test_loss_and_lora_grads_match_hf_oracle(load_in_4bit, head_dim):
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    model = _tiny(load_in_4bit, head_dim=head_dim)
    assert model.get_base_model()._unsloth_amd_patched == (2, 2, 2), "fast hooks not installed on every layer"
    ids, labels, pos = _batch()
    out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
    assert repr(out.logits) == "EMPTY_LOGITS"                       # fused CE never materialises logits
    out.loss.backward()
    ref_loss, ref_grads = hf_reference_loss_and_lora_grads(model, ids, labels, pos)
    assert abs(float(out.loss) - float(ref_loss)) <= 1e-2 * abs(float(ref_loss)), (float(out.loss), float(ref_loss))
    got = _grads(model)
    assert set(got) == set(ref_grads)
    worst = max(rel_fro(got[k], ref_grads[k]) for k in got)
    assert worst < 8e-2, worst
    total = rel_fro(torch.cat([got[k].flatten() for k in sorted(got)]),
                    torch.cat([ref_grads[k].flatten() for k in sorted(got)]))
    assert total < 3e-2, total


def test_gradient_checkpointing_is_bitwise_neutral_and_run_to_run_deterministic():
    ids, labels, pos = _batch(seed=1)
    res = []
    for gc in (True, False, True):
        model = _tiny(gc=gc)
        out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
        out.loss.backward()
        res.append((out.loss.detach().clone(), _grads(model)))
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], other[1][k]), k


def test_return_logits_branch_and_n_items():
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    model = _tiny()
    ids, labels, pos = _batch(seed=2)
    os.environ["UNSLOTH_RETURN_LOGITS"] = "1"
    try:
        out = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
    finally:
        os.environ.pop("UNSLOTH_RETURN_LOGITS")
    assert out.logits.shape == (2, 96, 1000)
    ref_loss, _ = hf_reference_loss_and_lora_grads(model, ids, labels, pos)
    assert abs(float(out.loss) - float(ref_loss)) <= 1e-2 * abs(float(ref_loss))
    # num_items_in_batch only rescales (global token count under DP)
    out2 = model(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV), num_items_in_batch=1000)
    n = int((labels[:, 1:] != -100).sum())
    assert abs(float(out2.loss) * 1000 / n - float(ref_loss)) <= 1e-2 * abs(float(ref_loss))


@pytest.mark.parametrize("head_dim", [32, 128])
def test_padding_free_packed_batch_equals_per_document_oracle(head_dim, monkeypatch):
    """position ids restart per document (indexed RoPE), attention is block-diagonal, boundary targets are
    masked: the packed row must equal the documents run separately. head_dim 128 = the band (lo, hi) path of the
    flash kernels, 32 = SDPA with the dense packed mask."""
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    from unsloth_amd.kernels import attention as flash
    from unsloth_amd.utils.packing import enable_padding_free_metadata
    bands = []
    real = flash.flash_attention
    monkeypatch.setattr(flash, "flash_attention", lambda q, k, v, s=None, band=None: (bands.append(band), real(q, k, v, s, band))[1])
    model = _tiny(head_dim=head_dim)
    g = torch.Generator().manual_seed(5)
    docs = [torch.randint(0, 1000, (n,), generator=g).tolist() for n in (40, 17, 64)]
    batch = enable_padding_free_metadata(docs, device=DEV)
    out = model(**batch)
    out.loss.backward()
    got = _grads(model)
    assert (len(bands) > 0 and all(b is not None for b in bands)) == (head_dim == 128)
    # oracle: each document alone; sum of token losses / total targets
    tot, n_tot, ref = 0.0, 0, None
    for d in docs:
        ids = torch.tensor([d])
        n = len(d) - 1
        loss, grads = hf_reference_loss_and_lora_grads(model, ids, ids.clone(), None)
        tot += float(loss) * n
        n_tot += n
        ref = {k: v * n for k, v in grads.items()} if ref is None else {k: ref[k] + grads[k] * n for k in ref}
    want = tot / n_tot
    assert abs(float(out.loss) - want) <= 1e-2 * abs(want), (float(out.loss), want)
    total = rel_fro(torch.cat([got[k].flatten() for k in sorted(got)]),
                    torch.cat([(ref[k] / n_tot).flatten() for k in sorted(got)]))
    assert total < 4e-2, total


def test_training_reduces_loss_and_adapters_round_trip(tmp_path):
    from unsloth_amd.trainer import make_optimizer, unsloth_train
    model = _tiny(r=16)
    ids, labels, pos = _batch(B=2, T=64, seed=3)
    batch = dict(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
    opt = make_optimizer(model, lr=2e-3)
    losses = unsloth_train(model, [batch] * 30, optimizer=opt)
    assert losses[-1] < 0.6 * losses[0], losses[::5]
    model.save_pretrained(str(tmp_path))
    from safetensors.torch import load_file
    sd = load_file(os.path.join(str(tmp_path), "adapter_model.safetensors"))
    assert any(k.endswith("q_proj.lora_A.weight") for k in sd) and len(sd) == 2 * 7 * 2
