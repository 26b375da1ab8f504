"""Host-side behaviour of optim.FlatAdamW / trainer.unsloth_train that needs no kernel launch: one gradient
arena per model, the step counter of a loaded state, checkpoints written before the first step."""
import pytest
import torch


def _model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16, bias=False), torch.nn.Linear(16, 4, bias=False))
    return m


def test_flat_adamw_refuses_a_second_arena_and_adopts_a_given_one():
    from unsloth_amd.dp import LoRAGradArena
    from unsloth_amd.optim import FlatAdamW
    m = _model()
    arena = LoRAGradArena(m)
    with pytest.raises(RuntimeError, match="already belong to a live"):
        FlatAdamW(m)
    opt = FlatAdamW(m, arena=arena)
    assert opt.arena is arena and not opt._owns_arena
    arena.close()


def test_unsloth_train_builds_the_arena_before_the_optimizer(monkeypatch):
    """world_size > 1: ONE LoRAGradArena, created first and handed to make_optimizer (trainer.py:159)."""
    from unsloth_amd import trainer as T
    made = []

    class FakeArena:
        def __init__(self, model):
            made.append(("arena", model))

        def finish(self):
            pass

    def fake_make_optimizer(model, arena=None, **kw):
        made.append(("optimizer", arena))
        return torch.optim.SGD(model.parameters(), lr=0.1)
    monkeypatch.setattr(T, "LoRAGradArena", FakeArena)
    monkeypatch.setattr(T, "make_optimizer", fake_make_optimizer)
    monkeypatch.setattr(torch.distributed, "is_initialized", lambda: True)
    monkeypatch.setattr(torch.distributed, "get_world_size", lambda *a, **k: 2)
    T.unsloth_train(_model(), batches=[])
    assert [k for k, _ in made] == ["arena", "optimizer"]
    assert isinstance(made[1][1], FakeArena), "the optimizer must be built over the data-parallel arena"


def test_load_state_dict_takes_the_loaded_step_and_tolerates_empty_state():
    from unsloth_amd.optim import FlatAdamW
    m = _model()
    opt = FlatAdamW(m)
    fresh = opt.state_dict()                       # saved before the first step
    opt._t = 7                                     # (as if 7 steps had run: no kernel needed for the bookkeeping)
    opt._step_t.fill_(7.0)
    late = opt.state_dict()
    for st in late["state"].values():
        st["step"] = torch.tensor(7.0)
    opt2 = FlatAdamW(_model())
    opt2._t = 20
    opt2.load_state_dict(late)
    assert opt2._t == 7 and float(opt2._step_t) == 7.0, "an earlier checkpoint must restart the bias correction there"
    empty = {"state": {}, "param_groups": fresh["param_groups"]}
    opt2.load_state_dict(empty)                    # must not raise KeyError('exp_avg')
    assert opt2._t == 0 and float(opt2.flat_m.abs().sum()) == 0.0
    for p, off, k, _ in opt2._views:
        assert opt2.state[p]["exp_avg"].data_ptr() == opt2.flat_m.data_ptr() + 4 * off
    opt.close(); opt2.close()


def test_exclude_rope_inv_freq_from_ddp_lists_the_rotary_buffers():
    """loader_utils.py:849-865 semantics: the HF rotary modules' inv_freq buffers end up in the list torch's DDP skips."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from unsloth_amd.dp import exclude_rope_inv_freq_from_ddp
    cfg = LlamaConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                      num_key_value_heads=1, vocab_size=50, max_position_embeddings=64)
    m = LlamaForCausalLM(cfg)
    m._ddp_params_and_buffers_to_ignore = ["something.else"]
    out = exclude_rope_inv_freq_from_ddp(m)
    assert out is m
    assert "model.rotary_emb.inv_freq" in m._ddp_params_and_buffers_to_ignore
    assert "something.else" in m._ddp_params_and_buffers_to_ignore
    exclude_rope_inv_freq_from_ddp(m)
    assert m._ddp_params_and_buffers_to_ignore.count("model.rotary_emb.inv_freq") == 1


def test_arena_dirtiness_counts_every_report_and_arrivals_once():
    """dp.LoRAGradArena on the CPU, one process: `writes` (what optim.FlatAdamW.zero_grad reads to decide whether the arena has to be
    cleared) grows with EVERY gradient report -- also the second one of a parameter (a sink's ready() followed by torch's
    post-accumulate hook, or a second micro-batch) --, while a bucket's arrival count takes each parameter once per accumulation.
    Round 4's first de-duplication skipped `writes` on the repeat: after a thrown-away backward, zero_grad() found the arena
    "clean" and the next accumulation started from the old sums (caught on the GPU by test_gradient_accumulation_over_micro_batches)."""
    import torch
    from unsloth_amd.dp import LoRAGradArena

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.zeros(4, 3))
            self.b = torch.nn.Parameter(torch.zeros(5))
    m = M()
    arena = LoRAGradArena(m, process_group=None, direct=False)
    assert arena.writes == 0 and len(arena.buckets) == 1
    for rep in range(2):                              # two backward passes without a reset in between (accumulation)
        for p in arena.params:
            arena.ready(p)                            # the sink's report
            arena.ready(p)                            # ... and torch's hook for the same gradient
    assert arena.writes == 8                          # every report counted
    assert len(arena._arrived) == 2 and arena._pending == [0]          # each parameter arrived once; the bucket completed once
    w = arena.writes
    arena.reset_arrivals()
    for p in arena.params:
        arena.ready(p)
    assert arena.writes == w + 2 and len(arena._arrived) == 2
    arena.close()
