"""Host logic of the selective-recompute policies (models/fast_layer.py): spec parsing, per-layer lookup and the
"unsloth:auto" schedule arithmetic -- the counterpart of the reference's smart-checkpointing selection
(unsloth/models/_utils.py:360-386), decided from free HBM instead of host-offload heuristics."""
import pytest

from unsloth_amd.models import fast_layer as F

GiB = 1 << 30
LLAMA3_8B = dict(n_layers=32, tokens=8192, hidden=4096, inter=14336, qkv_cols=6144, elsize=2, vocab=128256)


def test_policy_specs_and_layer_lookup():
    assert F.resolve_policy_spec("attn") == F.POLICIES["attn"]
    assert F.resolve_policy_spec("qkv+eg") == frozenset({"qkv", "eg"})
    sched = F.resolve_policy_spec("all*2,min*1,attn")
    assert [F.policy_for_layer(sched, i) for i in range(5)] == [F.POLICIES["all"]] * 2 + [F.POLICIES["min"]] + [F.POLICIES["attn"]] * 2
    assert F.resolve_policy_spec("auto") == F.AUTO and F.resolve_policy_spec(" auto ") == F.AUTO
    with pytest.raises(KeyError):
        F.resolve_policy_spec("everything")


def test_auto_schedule_is_monotone_and_bounded():
    ks = []
    for free in (0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 64, 250):
        pol = F.auto_schedule(free_bytes=free * GiB, **LLAMA3_8B)
        if pol == F.POLICIES["attn"]:
            k = 0
        elif pol == F.POLICIES["all"]:
            k = 32
        else:
            assert pol[0][1] == F.POLICIES["all"] and pol[1] == (None, F.POLICIES["attn"])
            k = pol[0][0]
            assert 0 < k < 32
            # the schedule is what policy_for_layer consumes
            assert F.policy_for_layer(pol, k - 1) == F.POLICIES["all"] and F.policy_for_layer(pol, k) == F.POLICIES["attn"]
        ks.append(k)
    assert ks == sorted(ks) and ks[0] == 0 and ks[-1] == 32
    # measured on the MI355X (profiles/r03final_bench.json): keep-everything peaks at 36.7 GB with 8.5 GB of weights and
    # optimizer state resident, "attn" at 18.7 GB -- so ~28 GB of free HBM must be enough for all 32 layers and ~10 GB must not
    assert F.auto_schedule(free_bytes=40 * GiB, **LLAMA3_8B) == F.POLICIES["all"]
    assert F.auto_schedule(free_bytes=10 * GiB, **LLAMA3_8B) == F.POLICIES["attn"]


def test_auto_schedule_never_plans_beyond_the_free_memory():
    per_tok_all = (2 * 4096 + 2 * 14336) * 2
    per_tok_attn = (4096 + 6144 + 4096 + 4096) * 2
    for free in (14, 18, 22, 26, 30):
        pol = F.auto_schedule(free_bytes=free * GiB, **LLAMA3_8B)
        k = 0 if pol == F.POLICIES["attn"] else 32 if pol == F.POLICIES["all"] else pol[0][0]
        kept = 8192 * (32 * per_tok_attn + k * per_tok_all)
        assert kept < free * GiB * 0.85, (free, k)
    # longer sequences / bigger batches shrink k at the same free memory
    a = F.auto_schedule(free_bytes=24 * GiB, **LLAMA3_8B)
    b = F.auto_schedule(free_bytes=24 * GiB, **dict(LLAMA3_8B, tokens=16384))
    ka = 32 if a == F.POLICIES["all"] else 0 if a == F.POLICIES["attn"] else a[0][0]
    kb = 32 if b == F.POLICIES["all"] else 0 if b == F.POLICIES["attn"] else b[0][0]
    assert kb < ka
