"""-m gpu: nf4.DecodeAhead on the device -- the NF4 decode of the next projection group runs on a side HIP stream while
the current GEMM runs. Training steps must be BITWISE equal to decoding in stream order (the decode is the same kernel
on the same bytes; only its stream changes), in every checkpointing mode, and the plan must actually be followed
(hits counted). The scheduling logic itself is covered on CPU by tests/test_decode_ahead.py."""
import pytest
import torch

from tests.test_gpu_model import DEV, _batch, _grads, _tiny

pytestmark = pytest.mark.gpu


def _run(gc, ahead, steps=4, eval_between=False):
    from unsloth_amd import nf4
    from unsloth_amd.kernels import utils as U
    ids, labels, pos = _batch(B=2, T=160, seed=11)
    batch = dict(input_ids=ids.to(DEV), labels=labels.to(DEV), position_ids=pos.to(DEV))
    saved = (U.FUSED_NF4, U.GEMM256_MODE, nf4.DECODE_AHEAD)
    U.FUSED_NF4, U.GEMM256_MODE = False, "on"       # the decode-once + 256-tile path of the benchmark sizes
    nf4.set_decode_ahead(False)                     # drops any plan a previous test left behind
    nf4.set_decode_ahead(ahead)
    try:
        model = _tiny(gc=gc, head_dim=128)
        losses = []
        for step in range(steps):
            for p_ in model.parameters():
                p_.grad = None
            out = model(**batch)
            out.loss.backward()
            losses.append(out.loss.detach().clone())
            if eval_between and step == 1:
                with torch.no_grad():
                    model(**batch)
        torch.cuda.synchronize()
        stats = nf4.decode_ahead_stats().get(0, {})
        return torch.stack(losses), _grads(model), dict(stats)
    finally:
        U.FUSED_NF4, U.GEMM256_MODE = saved[0], saved[1]
        nf4.set_decode_ahead(saved[2])


@pytest.mark.parametrize("gc", [False, "unsloth", True])
def test_decode_ahead_is_bitwise_neutral(gc):
    from unsloth_amd import nf4
    base_loss, base_grads, s0 = _run(gc, False)
    h0 = s0.get("hits", 0)
    loss, grads, s1 = _run(gc, True)
    assert torch.equal(base_loss, loss), (base_loss, loss)
    for k in base_grads:
        assert torch.equal(base_grads[k], grads[k]), k
    # 2 layers x (4 forward + 5 backward requests [+ recomputed forwards]); steps 2..4 follow the plan
    assert s1["hits"] - h0 >= 3 * (2 * 9 - 1), s1
    assert s1["plan"] >= 2 * 9


def test_decode_ahead_survives_a_different_call_sequence():
    base_loss, base_grads, _ = _run(False, False, eval_between=True)
    loss, grads, s = _run(False, True, eval_between=True)
    assert torch.equal(base_loss, loss)
    for k in base_grads:
        assert torch.equal(base_grads[k], grads[k]), k
    assert s["misses"] > 0 and s["hits"] > 0
