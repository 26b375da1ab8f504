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


def test_bare_unsloth_spelling_is_the_fit_to_memory_schedule(monkeypatch):
    """`use_gradient_checkpointing="unsloth"` (the API default, reference models/loader.py:407-441) resolves to the
    least-recompute schedule that fits (the reference's own "unsloth" is a fit-to-memory decision, models/_utils.py:360-386);
    the fixed keep-attention policy is "unsloth:attn"; UNSLOTH_AMD_GC_POLICY re-binds the bare spelling."""
    import torch
    from unsloth_amd.models.llama import FastLlamaModel

    class _Inner(torch.nn.Module):
        gradient_checkpointing = False

    class _Base(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = _Inner()

    def policy_of(mode):
        b = _Base()
        FastLlamaModel.for_training(b, mode)
        return b.model._unsloth_amd_layer_policy, b.model.gradient_checkpointing

    monkeypatch.delenv("UNSLOTH_AMD_GC_POLICY", raising=False)
    assert policy_of("unsloth") == (F.AUTO, True)
    assert policy_of("unsloth:auto") == (F.AUTO, True)
    assert policy_of("unsloth:attn") == (F.POLICIES["attn"], True)
    assert policy_of("unsloth:min") == (F.POLICIES["min"], True)
    assert policy_of(True) == (None, True) and policy_of(False) == (None, False)
    monkeypatch.setenv("UNSLOTH_AMD_GC_POLICY", "attn")
    assert policy_of("unsloth") == (F.POLICIES["attn"], True)
    assert policy_of("unsloth:all") == (F.POLICIES["all"], True)      # an explicit name beats the environment


def test_auto_schedule_on_an_idle_and_on_a_crowded_mi355x():
    """The two operating points bench.py reports for the default spelling: an idle 288 GB part keeps everything in all 32
    layers (the speed of gradient_checkpointing=False); with 10 GB free beyond the 8.5 GB of resident weights and optimizer
    state (a 19 GB total footprint, what the fixed "attn" policy measures) it is the keep-attention policy everywhere."""
    assert F.auto_schedule(free_bytes=270 * GiB, **LLAMA3_8B) == F.POLICIES["all"]
    assert F.auto_schedule(free_bytes=10 * GiB, **LLAMA3_8B) == F.POLICIES["attn"]
    # batch 1 (2048 tokens) needs a quarter of the activations: the same 10 GB are plenty for keep-everything
    assert F.auto_schedule(free_bytes=10 * GiB, **dict(LLAMA3_8B, tokens=2048)) == F.POLICIES["all"]


def test_decoded_weight_mirrors_ride_on_the_same_decision():
    """nf4.RESIDENT_MODE "auto": the bare "unsloth" spelling also keeps 16-bit mirrors of the NF4 projections when every layer keeps
    everything AND the HBM left after that holds the mirrors twice over (turning on), or the step still fits with them (staying
    on). Llama-3-8B, 8192 tokens: "all" takes ~28 GB beyond weights and optimizer, the mirrors 14 GB."""
    from unsloth_amd.models import fast_layer as F
    kw = dict(n_layers=32, tokens=8192, hidden=4096, inter=14336, qkv_cols=6144, elsize=2, vocab=128256)
    G = 1 << 30
    assert F.mirrors_fit(free_bytes=250 * G, **kw)                       # an idle MI355X
    assert F.mirrors_fit(free_bytes=80 * G, **kw)
    assert not F.mirrors_fit(free_bytes=48 * G, **kw)                    # "all" fits (auto_schedule), twice 14 GB on top does not
    assert F.auto_schedule(free_bytes=48 * G, **kw) == F.POLICIES["all"]
    assert F.mirrors_fit(free_bytes=40 * G, have_mirrors=True, **kw)     # already allocated: stay while "all" still fits
    assert not F.mirrors_fit(free_bytes=20 * G, have_mirrors=True, **kw)
    # tiny batches on a big model: the mirrors are what a short step gains most from (9 % at 2048 tokens) and fit easily
    assert F.mirrors_fit(**dict(kw, tokens=2048), free_bytes=60 * G)
