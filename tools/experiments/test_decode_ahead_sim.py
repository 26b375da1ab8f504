"""nf4.DecodeAhead (the NF4 decode of the next projection group on a side stream while the current GEMM runs):
the scheduling logic against a two-stream SIMULATOR on CPU. The simulator executes the enqueued operations under
random interleavings that respect stream order and event waits, gives every decode and every consumer a duration
(begin / end), and flags (a) a consumer that reads anything but the decode of exactly the weights it asked for,
(b) a write that begins while a read or another write of the same slot is in progress. No GPU, no kernels: the HIP
side of the same class is covered by tests/test_gpu_decode_ahead.py (bitwise-equal training steps)."""
import random

import pytest
import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import decode_ahead as nf4


class _Stream:
    def __init__(self, name):
        self.name, self.q = name, []


class _Event:
    def __init__(self):
        self.done = False


class _Weight:
    """Stands for a packed tensor AND its quant state (weakref-able, shape/dtype like nf4.QuantState)."""
    _next = [1]

    def __init__(self, rows, cols=8, dtype=torch.float32):   # fp32: the tags written into the buffers stay exact
        self.shape, self.dtype, self.nested, self._absmax_f32 = (rows, cols), dtype, True, object()
        self.tag = _Weight._next[0]
        _Weight._next[0] += 1


class SimBackend:
    def __init__(self, rng):
        self.rng = rng
        self._main, self.side = _Stream("main"), _Stream("side")
        self.errors = []
        self.busy = {}                 # id(storage) -> {"w": n writers, "r": n readers}
        self.allocs = 0

    # ---- what DecodeAhead calls
    def main(self):
        return self._main

    def record(self, stream):
        e = _Event()
        stream.q.append(("record", e))
        return e

    def wait(self, stream, event):
        stream.q.append(("wait", event))

    def alloc(self, numel, dtype):
        self.allocs += 1
        return torch.full((numel,), -1.0, dtype=torch.float32).to(dtype)

    def release(self, buf):
        self.drain()

    def decode(self, stream, packed_list, qs_list, buf2d):
        tags = [q.tag for q in qs_list]
        rows = [q.shape[0] for q in qs_list]
        stream.q.append(("wbegin", buf2d))
        stream.q.append(("wend", buf2d, tags, rows))

    def decodable_off_stream(self, qs_list):
        return all(q._absmax_f32 is not None for q in qs_list)

    # ---- what the test harness calls
    def consume(self, views, qs_list):
        """The GEMM that follows a fetch: reads the views over a duration."""
        self._main.q.append(("rbegin", views, [q.tag for q in qs_list]))
        self._main.q.append(("rend", views, [q.tag for q in qs_list]))

    def _state(self, t):
        return self.busy.setdefault(t.untyped_storage().data_ptr(), {"w": 0, "r": 0})

    def _check(self, views, tags, when):
        for v, tag in zip(views, tags):
            if not bool((v.float() == float(tag)).all()):
                self.errors.append(f"{when}: consumer of weight {tag} read {v.float().flatten()[:4].tolist()}")

    def _run(self, stream):
        op = stream.q[0]
        kind = op[0]
        if kind == "wait":
            if not op[1].done:
                return False
        elif kind == "record":
            op[1].done = True
        elif kind == "wbegin":
            st = self._state(op[1])
            if st["w"] or st["r"]:
                self.errors.append(f"write begins on {stream.name} while {st} in progress")
            st["w"] += 1
            op[1].fill_(-7.0)                     # a half-written buffer holds garbage
        elif kind == "wend":
            _, buf2d, tags, rows = op
            r = 0
            for tag, n in zip(tags, rows):
                buf2d[r:r + n] = float(tag)
                r += n
            self._state(buf2d)["w"] -= 1
        elif kind == "rbegin":
            st = self._state(op[1][0])
            if st["w"]:
                self.errors.append("read begins while a write is in progress")
            st["r"] += 1
            self._check(op[1], op[2], "rbegin")
        elif kind == "rend":
            self._check(op[1], op[2], "rend")
            self._state(op[1][0])["r"] -= 1
        stream.q.pop(0)
        return True

    def drain(self):
        streams = [self._main, self.side]
        while any(s.q for s in streams):
            order = [s for s in streams if s.q]
            self.rng.shuffle(order)
            # bias: sometimes let one stream run far ahead
            burst = self.rng.choice([1, 1, 2, 5, 50])
            progressed = False
            for s in order:
                for _ in range(burst):
                    if not s.q or not self._run(s):
                        break
                    progressed = True
            assert progressed, "deadlock: both streams wait on events that nobody records"


def _model(n_layers):
    """Weights of a tiny decoder: per layer q|k|v, o, gate|up, down."""
    return [dict(qkv=[_Weight(4), _Weight(2), _Weight(2)], o=[_Weight(4)], gu=[_Weight(6), _Weight(6)], d=[_Weight(4)])
            for _ in range(n_layers)]


def _train_step(da, be, layers):
    """The request sequence of kernels/utils.py for forward + backward without checkpointing."""
    da.step_begin()
    seq = []
    for L in layers:                                                   # forward
        seq += [L["qkv"], L["o"], L["gu"], L["d"]]
    for L in reversed(layers):                                         # backward: down, gate, up (two launches), o, q|k|v merged
        seq += [L["d"], [L["gu"][0]], [L["gu"][1]], L["o"], L["qkv"]]
    for group in seq:
        buf, views = da.fetch(group, group)
        assert buf.shape[0] == sum(w.shape[0] for w in group)
        be.consume(views, group)
    return len(seq)


def _eval_step(da, be, layers):
    da.step_begin()
    for L in layers:
        for group in (L["qkv"], L["o"], L["gu"], L["d"]):
            _, views = da.fetch(group, group)
            be.consume(views, group)


@pytest.mark.parametrize("seed", range(25))
def test_steady_state_hits_and_no_hazard(seed):
    rng = random.Random(seed)
    be = SimBackend(rng)
    da = nf4.DecodeAhead(be)
    layers = _model(3)
    n = _train_step(da, be, layers)
    assert (da.hits, da.misses) == (0, n)                  # first step: recorded, decoded in stream order
    for _ in range(3):
        _train_step(da, be, layers)
    be.drain()
    assert not be.errors, be.errors[:3]
    # every later step: all but its first request were decoded ahead (no prefetch crosses a step boundary)
    assert da.misses == n + 3 and da.hits == 3 * (n - 1)
    assert be.allocs <= 4                                  # two slots, each grown at most once


@pytest.mark.parametrize("seed", range(25))
def test_misprediction_falls_back_in_stream_order(seed):
    rng = random.Random(1000 + seed)
    be = SimBackend(rng)
    da = nf4.DecodeAhead(be)
    layers = _model(2)
    _train_step(da, be, layers)
    _train_step(da, be, layers)
    _eval_step(da, be, layers)          # follows the plan through the forward, then the plan's backward never comes
    _eval_step(da, be, layers)
    _train_step(da, be, layers)         # plan = an eval step: mispredicts at the first backward request
    other = _model(2)
    _train_step(da, be, other)          # another model entirely
    _train_step(da, be, other)
    be.drain()
    assert not be.errors, be.errors[:3]
    assert da.hits > 0 and da.misses > 0


def test_dead_weights_and_growth():
    rng = random.Random(7)
    be = SimBackend(rng)
    da = nf4.DecodeAhead(be)
    layers = _model(2)
    _train_step(da, be, layers)
    _train_step(da, be, layers)
    del layers[1]["o"][0]               # the plan only holds weak references
    import gc
    gc.collect()
    layers[1]["o"] = [_Weight(4)]
    _train_step(da, be, layers)
    big = [dict(qkv=[_Weight(40), _Weight(20), _Weight(20)], o=[_Weight(40)], gu=[_Weight(60), _Weight(60)], d=[_Weight(40)])]
    _train_step(da, be, big)            # larger than both slots: grown behind a device sync
    _train_step(da, be, big)
    be.drain()
    assert not be.errors, be.errors[:3]


def test_no_prefetch_without_cached_absmax():
    be = SimBackend(random.Random(3))
    da = nf4.DecodeAhead(be)
    layers = _model(1)
    _train_step(da, be, layers)
    for L in layers:
        for g in L.values():
            for w in g:
                w._absmax_f32 = None
    _train_step(da, be, layers)
    be.drain()
    assert da.prefetches == 0 and not be.errors


@pytest.mark.parametrize("drop", ["gate", "main_wait"])
def test_simulator_detects_a_missing_dependency(drop):
    """The simulator is only worth something if it catches a broken scheduler: drop (a) the event that orders the
    side-stream overwrite behind the slot's last consumer, (b) the main stream's wait for the decode it is handed."""
    class Broken(SimBackend):
        def wait(self, stream, event):
            if (drop == "gate" and stream is self.side) or (drop == "main_wait" and stream is self._main):
                return
            super().wait(stream, event)

    caught = 0
    for seed in range(10):
        be = Broken(random.Random(seed))
        da = nf4.DecodeAhead(be)
        layers = _model(3)
        for _ in range(3):
            _train_step(da, be, layers)
        be.drain()
        caught += bool(be.errors)
    assert caught == 10
