"""GRPO / DPO drivers (models/rl_replacements.py). Host side here: the packed-forward index bookkeeping
(integer work, exact) and the loss formulas against direct restatements of TRL's; the GPU leg is tests/test_gpu_rl_drivers.py."""
import math

import pytest
import torch


def _rows(gen, B=3, P=6, C=5, vocab=50, pad=0):
    """left-padded prompts | right-padded completions, like TRL's GRPO batches."""
    ids = torch.full((B, P + C), pad)
    mask = torch.zeros(B, P + C, dtype=torch.long)
    plen = [6, 3, 4][:B]
    clen = [5, 2, 4][:B]
    for b in range(B):
        ids[b, P - plen[b]:P] = torch.randint(1, vocab, (plen[b],), generator=gen)
        ids[b, P:P + clen[b]] = torch.randint(1, vocab, (clen[b],), generator=gen)
        mask[b, P - plen[b]:P + clen[b]] = 1
    return ids, mask, plen, clen


def test_packed_completion_index_is_exact():
    from unsloth_amd.models.rl_replacements import _packed_completion_index
    gen = torch.Generator().manual_seed(0)
    ids, mask, plen, clen = _rows(gen)
    P, C = 6, 5
    flat, pos, lens, src, tgt, (r, c) = _packed_completion_index(ids, mask, C)
    assert lens.tolist() == [p + q for p, q in zip(plen, clen)] and lens.dtype == torch.int32
    want_flat = torch.cat([ids[b][mask[b].bool()] for b in range(3)])
    assert torch.equal(flat[0], want_flat)
    assert torch.equal(pos[0], torch.cat([torch.arange(n) for n in lens.tolist()]))
    # every completion token of every row appears exactly once as a target, predicted from the token before it
    want = [(b, j) for b in range(3) for j in range(clen[b])]
    assert list(zip(r.tolist(), c.tolist())) == want
    off = [0] + torch.cumsum(lens, 0).tolist()
    for (b, j), s, t in zip(want, src.tolist(), tgt.tolist()):
        assert s == off[b] + plen[b] + j - 1 and t == int(ids[b, P + j])
    # no mask: every column is a token
    flat2, _, lens2, src2, _, _ = _packed_completion_index(ids, None, C)
    assert lens2.tolist() == [P + C] * 3 and src2.numel() == 3 * C


@pytest.mark.parametrize("loss_type", ["grpo", "bnpo", "dr_grpo", "dapo"])
@pytest.mark.parametrize("beta,delta,level", [(0.0, None, "token"), (0.04, None, "token"), (0.04, 1.5, "sequence")])
def test_grpo_loss_matches_trl_formula(loss_type, beta, delta, level):
    from unsloth_amd.models.rl_replacements import grpo_compute_loss
    g = torch.Generator().manual_seed(1)
    B, L = 4, 7
    new = (torch.randn(B, L, generator=g) * 0.3 - 2).requires_grad_(True)
    old = new.detach() + torch.randn(B, L, generator=g) * 0.2
    ref = new.detach() + torch.randn(B, L, generator=g) * 0.1
    mask = (torch.rand(B, L, generator=g) > 0.25).long()
    adv = torch.randn(B, generator=g)
    loss, length, kl, coef, _ = grpo_compute_loss(ref, new, old, mask, adv, beta=beta, loss_type=loss_type, epsilon_low=0.2,
                                                  epsilon_high=0.28, delta=delta, max_completion_length=L,
                                                  num_items_in_batch=torch.tensor(19.0), num_processes=1,
                                                  importance_sampling_level=level)
    # straight restatement of trl/trainer/grpo_trainer.py::_compute_loss
    m = mask.float()
    lr = new - old
    lw = lr if level == "token" else ((lr * m).sum(-1) / m.sum(-1).clamp(min=1)).unsqueeze(-1)
    c1 = torch.exp(lw)
    c2 = torch.clamp(c1, 0.8, 1.28)
    if delta is not None:
        c1 = torch.clamp(c1, max=delta)
    ptl = -torch.min(c1 * adv[:, None], c2 * adv[:, None])
    if beta:
        d = ref - new
        ptl = ptl + beta * (torch.exp(d) - d - 1)
    want = {"grpo": ((ptl * m).sum(-1) / m.sum(-1).clamp(min=1)).mean(), "bnpo": (ptl * m).sum() / m.sum().clamp(min=1),
            "dr_grpo": (ptl * m).sum() / (B * L), "dapo": (ptl * m).sum() / 19.0}[loss_type]
    assert torch.allclose(loss, want, rtol=1e-6, atol=1e-7)
    (gw,) = torch.autograd.grad(want, new, retain_graph=True)
    (gg,) = torch.autograd.grad(loss, new)
    assert torch.allclose(gg, gw, rtol=1e-5, atol=1e-7)
    assert abs(float(length) - float(m.sum(-1).mean())) < 1e-6


def test_dpo_loss_matches_formula():
    from unsloth_amd.models.rl_replacements import dpo_loss
    g = torch.Generator().manual_seed(2)
    pc, pr, rc, rr = (torch.randn(5, generator=g) * 3 - 20 for _ in range(4))
    losses, cw, rw = dpo_loss(pc, pr, rc, rr, beta=0.1)
    z = 0.1 * ((pc - pr) - (rc - rr))
    assert torch.allclose(losses, torch.log1p(torch.exp(-z)), rtol=1e-5, atol=1e-6)
    assert torch.allclose(cw, 0.1 * (pc - rc)) and torch.allclose(rw, 0.1 * (pr - rr))
    ls, _, _ = dpo_loss(pc, pr, rc, rr, beta=0.1, label_smoothing=0.1)
    assert torch.allclose(ls, 0.9 * torch.log1p(torch.exp(-z)) + 0.1 * torch.log1p(torch.exp(z)), rtol=1e-5, atol=1e-6)
    li, _, _ = dpo_loss(pc, pr, rc, rr, beta=0.1, loss_type="ipo")
    assert torch.allclose(li, ((pc - pr) - (rc - rr) - 5.0) ** 2)
    with pytest.raises(ValueError):
        dpo_loss(pc, pr, rc, rr, loss_type="nope")
