"""CPU: the decode engine's logits filter (models/decode.py filter_logits) against HuggingFace's own warpers
(transformers/generation/logits_process.py TopKLogitsWarper -> TopPLogitsWarper, the order `generate` applies them in)."""
import pytest
import torch


@pytest.mark.parametrize("top_k,top_p", [(0, 0.9), (50, 1.0), (20, 0.8), (0, 0.05), (5, 0.999), (0, 1.0)])
def test_filter_matches_hf_warpers(top_k, top_p):
    from transformers.generation.logits_process import TopKLogitsWarper, TopPLogitsWarper
    from unsloth_amd.models.decode import filter_logits
    g = torch.Generator().manual_seed(top_k * 7 + int(top_p * 100))
    logits = torch.randn(4, 1000, generator=g) * 3.0
    logits[1, :10] = logits[1, 10]                   # ties
    want = logits.clone()
    if top_k:
        want = TopKLogitsWarper(top_k=top_k)(None, want)
    if top_p < 1.0:
        want = TopPLogitsWarper(top_p=top_p)(None, want)
    got = filter_logits(logits.clone(), top_k, top_p)
    assert torch.equal(torch.isinf(got), torch.isinf(want))
    assert torch.equal(got[~torch.isinf(got)], want[~torch.isinf(want)])
    assert bool((~torch.isinf(got)).any(dim=-1).all())             # the most likely token always survives
