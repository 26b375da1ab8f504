"""-m gpu: chunked per-token log-probs (GRPO/DPO path) against the plain fp32 formula
log_softmax(f(h @ W^T))[index] on the CPU. unsloth_zoo's implementation is not in the repository (parity unpinned);
tolerance: 2e-3 of the log-prob scale (the logits chunk is rounded to bf16 by the GEMM -- the reference's own rounding
point: zoo computes the chunk under bf16 autocast -- so |err| ~ 2^-9 |logit|)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_logprobs(h, W, idx, mult=0.0, div=0.0, cap=0.0, temp=1.0):
    logits = h.float() @ W.float().t()
    if mult:
        logits = logits * mult
    if div:
        logits = logits / div
    if cap:
        logits = cap * torch.tanh(logits / cap)
    if temp != 1.0:
        logits = logits / temp
    return torch.log_softmax(logits, dim=-1).gather(-1, idx.unsqueeze(-1)).squeeze(-1)


@pytest.mark.parametrize("B,L,H,V,kw", [
    (2, 96, 256, 1000, {}),
    (1, 300, 512, 32000, dict(temperature=0.7)),
    (3, 65, 256, 5000, dict(logit_scale_multiply=0.5, logit_scale_divide=2.0)),
    (2, 64, 256, 70000, dict(logit_softcapping=30.0)),             # vocab > 65536: the CE kernel has no chunk limit
    (1, 513, 256, 1000, dict(chunks=3, temperature=1.3)),
    (2, 40, 256, 32001, {}),                                        # vocab % 8 != 0
])
def test_hidden_states_logprobs_forward_backward(B, L, H, V, kw):
    from unsloth_amd.models.rl_replacements import chunked_hidden_states_selective_log_softmax as f
    g = torch.Generator().manual_seed(0)
    h = (torch.randn(B, L, H, generator=g) * 0.5).to(torch.bfloat16)
    W = (torch.randn(V, H, generator=g) * 0.1).to(torch.bfloat16)
    idx = torch.randint(0, V, (B, L), generator=g)
    up = torch.randn(B, L, generator=g)
    hr = h.float().requires_grad_(True)
    want = ref_logprobs(hr, W, idx, kw.get("logit_scale_multiply", 0.0), kw.get("logit_scale_divide", 0.0),
                        kw.get("logit_softcapping", 0.0), kw.get("temperature", 1.0))
    (want * up).sum().backward()
    hd = h.to(DEV).requires_grad_(True)
    got = f(hd, W.to(DEV), idx.to(DEV), **kw)
    assert got.shape == (B, L) and got.dtype == torch.float32
    scale = want.detach().abs().max().item() + 1.0
    assert (got.detach().cpu() - want.detach()).abs().max().item() <= 2e-3 * scale
    (got * up.to(DEV)).sum().backward()
    gref = hr.grad
    err = (hd.grad.float().cpu() - gref).norm() / gref.norm()
    assert err <= 1.5e-2, float(err)


def test_logits_logprobs_match_formula():
    from unsloth_amd.models.rl_replacements import chunked_selective_log_softmax
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn(2, 50, 3000, generator=g) * 3).to(torch.bfloat16)
    idx = torch.randint(0, 3000, (2, 50), generator=g)
    want = torch.log_softmax(logits.float() / 0.9, -1).gather(-1, idx.unsqueeze(-1)).squeeze(-1)
    got = chunked_selective_log_softmax(logits.to(DEV), idx.to(DEV), temperature=0.9)
    assert (got.cpu() - want).abs().max().item() <= 1e-3 * (want.abs().max().item() + 1)


def test_softcap_with_temperature_is_refused():
    from unsloth_amd.models.rl_replacements import chunked_hidden_states_selective_log_softmax as f
    h = torch.zeros(1, 8, 64, device=DEV, dtype=torch.bfloat16)
    W = torch.zeros(100, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        f(h, W, torch.zeros(1, 8, dtype=torch.long, device=DEV), logit_softcapping=30.0, temperature=0.5)
