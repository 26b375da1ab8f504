"""-m gpu: the GRPO / DPO drivers end to end on the HIP path against an fp32 stock-HF model (oracle/ref_model.py): per-token
log-probs of left-padded prompt | right-padded completion rows through ONE packed forward (band attention, restarting
positions, lm_head on completion positions only), the GRPO objective and its LoRA gradients, the trainer-method wrappers on
a stand-in trainer object, the DPO sequence log-probs."""
import types

import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _model():
    from tests.test_gpu_model import _tiny
    return _tiny(load_in_4bit=True, gc=False, head_dim=128, r=8)


def _batch(gen, B=4, P=24, C=40, vocab=1000, pad=0):
    ids = torch.full((B, P + C), pad)
    mask = torch.zeros(B, P + C, dtype=torch.long)
    plen, clen = [24, 7, 15, 20], [40, 11, 33, 1]
    for b in range(B):
        ids[b, P - plen[b]:P] = torch.randint(1, vocab, (plen[b],), generator=gen)
        ids[b, P:P + clen[b]] = torch.randint(1, vocab, (clen[b],), generator=gen)
        mask[b, P - plen[b]:P + clen[b]] = 1
    return ids, mask, plen, clen


def _oracle_logps(model, ids, mask, C, temperature, objective=None):
    """fp32 stock HF, every row alone without its padding: log_softmax(logits / temperature)[next token] on the completion
    columns; optionally a scalar objective of them with LoRA gradients."""
    from oracle.ref_model import hf_reference_loss_and_lora_grads
    B, T = ids.shape
    out = torch.zeros(B, C)
    grads = None
    for b in range(B):
        keep = mask[b].bool()
        row = ids[b][keep].unsqueeze(0)
        n = row.shape[1]
        cols = keep.nonzero().squeeze(1)
        box = {}

        def fn(logits, row=row, n=n, cols=cols, b=b):
            lp = torch.log_softmax(logits[0, :-1].float() / temperature, dim=-1).gather(-1, row[0, 1:].to(logits.device).unsqueeze(-1)).squeeze(-1)
            full = torch.zeros(C, device=lp.device)
            tcol = cols[1:] - (T - C)
            sel = tcol >= 0
            full = full.index_put((tcol[sel].to(lp.device),), lp[sel.to(lp.device)])
            box["lp"] = full.detach().cpu()
            return objective(full, b) if objective is not None else full.sum()
        _, g = hf_reference_loss_and_lora_grads(model, row, row.clone(), torch.arange(n).unsqueeze(0), loss_fn=fn)
        out[b] = box["lp"]
        if objective is not None:
            grads = g if grads is None else {k: grads[k] + g[k] for k in g}
    return out, grads


def test_per_token_logps_of_padded_rows_match_fp32_oracle():
    from unsloth_amd.models.rl_replacements import dpo_sequence_logps, get_per_token_logps_and_entropies
    model = _model()
    gen = torch.Generator().manual_seed(7)
    ids, mask, plen, clen = _batch(gen)
    C = 40
    want, _ = _oracle_logps(model, ids, mask, C, 0.9)
    got, ent = get_per_token_logps_and_entropies(model, ids.to(DEV), mask.to(DEV), C, temperature=0.9, compute_entropy=True)
    assert got.shape == (4, C) and got.dtype == torch.float32
    cm = mask[:, -C:].bool()
    assert float(got.cpu()[~cm].abs().max()) == 0.0                       # padding columns stay 0
    scale = want.abs().max().item() + 1.0
    assert (got.cpu() - want)[cm].abs().max().item() <= 2e-3 * scale
    assert ent is not None and bool((ent[cm.to(DEV)] > 0).all()) and float(ent.max()) <= 6.91 + 1e-3     # <= ln(1000)
    seq = dpo_sequence_logps(model, ids.to(DEV), mask.to(DEV), mask[:, -C:].to(DEV), temperature=0.9)
    assert torch.allclose(seq.cpu(), (want * cm).sum(-1), rtol=2e-3, atol=2e-2)


def test_grpo_objective_and_lora_gradients_match_fp32_oracle():
    from unsloth_amd.models.rl_replacements import grpo_trainer_compute_loss
    model = _model()
    gen = torch.Generator().manual_seed(8)
    ids, mask, plen, clen = _batch(gen)
    P, C = 24, 40
    adv = torch.tensor([0.7, -1.1, 0.3, 0.5])
    old = torch.randn(4, C, generator=gen) * 0.05                         # old = oracle's new + noise (set below)
    lp0, _ = _oracle_logps(model, ids, mask, C, 1.0)
    old_lp, ref_lp = lp0 + old, lp0 - 0.5 * old
    cm = mask[:, -C:].float()

    def objective(full, b):                                              # this row's share of the "grpo" aggregation
        r = torch.exp(full - old_lp[b].to(full.device))
        c2 = torch.clamp(r, 0.8, 1.2)
        a = adv[b].item()
        ptl = -torch.min(r * a, c2 * a)
        d = ref_lp[b].to(full.device) - full
        ptl = ptl + 0.04 * (torch.exp(d) - d - 1)
        m = cm[b].to(full.device)
        return (ptl * m).sum() / m.sum().clamp(min=1) / 4.0

    _, ref_grads = _oracle_logps(model, ids, mask, C, 1.0, objective)
    want = sum(float(objective(lp0[b], b)) for b in range(4))
    trainer = types.SimpleNamespace(beta=0.04, epsilon_low=0.2, epsilon_high=0.2, temperature=1.0,
                                    importance_sampling_level="token",
                                    args=types.SimpleNamespace(loss_type="grpo", delta=None, max_completion_length=C),
                                    accelerator=types.SimpleNamespace(num_processes=1),
                                    _metrics={"train": {"completion_length": [], "kl": [], "clip_ratio/region_mean": []}},
                                    control=types.SimpleNamespace(should_evaluate=False))
    inputs = dict(prompt_ids=ids[:, :P].to(DEV), prompt_mask=mask[:, :P].to(DEV), completion_ids=ids[:, P:].to(DEV),
                  completion_mask=mask[:, P:].to(DEV), advantages=adv.to(DEV), old_per_token_logps=old_lp.to(DEV),
                  ref_per_token_logps=ref_lp.to(DEV))
    loss = grpo_trainer_compute_loss(trainer, model, inputs)
    assert abs(float(loss) - want) <= 2e-3 * max(1.0, abs(want)), (float(loss), want)
    loss.backward()
    got = {"layers." + n.split(".layers.", 1)[1].replace(".default.weight", ""): p.grad.detach().float().cpu()
           for n, p in model.named_parameters() if p.requires_grad}
    total = rel_fro(torch.cat([got[k].flatten() for k in sorted(got)]), torch.cat([ref_grads[k].flatten() for k in sorted(got)]))
    assert total < 3e-2, total
    assert trainer._metrics["train"]["completion_length"] and trainer._metrics["train"]["kl"][0] >= 0.0
    with pytest.raises(ValueError):
        grpo_trainer_compute_loss(trainer, model, inputs, return_outputs=True)
