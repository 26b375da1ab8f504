"""EXPERIMENT (round 2, measured slower -> not in the product): NF4 decode of the NEXT projection group on a side HIP
stream while the current GEMM runs. This is the scheduler as it was wired into unsloth_amd/nf4.py (decode_group /
decode_ahead_step_begin, called from kernels/utils.py and models/llama.py) at commit "nf4.DecodeAhead ...".

Result on MI355X (profiles/r02p_decode_ahead_ab.json, same process, same box, Llama-3-8B QLoRA):
    4 x 2048 tokens: 250.7 ms/step WITH the overlap vs 245.5 ms in stream order (GEMM 1.41 vs 1.48 PFLOP/s in-step)
    1 x 2048 tokens:  89.2 ms vs 83.9 ms (GEMM 1.12 vs 1.29 PFLOP/s)
The training steps were bitwise identical and the plan was followed (all but the first request of a step decoded
ahead) -- the schedule works, the hardware does not reward it: the GEMM launches got slower by MORE than the whole
decode time they were supposed to hide (+28 us per launch x 292 = +8.2 ms against 7.5 ms of decode). The GEMM runs at a
power-limited clock and feeds its LDS-DMA from L2 / Infinity Cache; a co-running kernel that streams 0.3 GB through the
same caches at HBM speed costs it more than running that kernel alone does. Consistent with the persistent-walk result
(DESIGN 5.6): on this device the currency is energy / bytes moved per step, not idle issue slots -- keeping the decoded
weights resident (nf4.set_resident, +14 GB) is what removes the decode cost (+2.9 % / +7.5 %).

tests/ of the experiment: `python -m pytest tools/experiments/test_decode_ahead_sim.py` (two-stream simulator, CPU).
"""
import os as _os
import weakref as _weakref

import torch

DECODE_AHEAD = False


def dequantize_nf4(*a, **k):          # the product's decode launch (unsloth_amd.nf4.dequantize_nf4) when run on a GPU
    from unsloth_amd.nf4 import dequantize_nf4 as f
    return f(*a, **k)


# ------------------------------------------------------------------------------------------------
# Decode-ahead: the NF4 decode of the NEXT projection group runs on a side HIP stream while the MFMA GEMM of the
# current one owns the matrix pipe.
#
# Why: the decode-once policy (kernels/utils.py) writes every weight as bf16 twice per step -- 224 launches, 7.5 ms of
# a 250 ms step at 8192 tokens and 9 % of the step at 2048 -- and that kernel is pure HBM streaming (20 VGPRs, 1 KiB of
# LDS) while the GEMM it sits in front of is MFMA-bound with half the register file and 32 KiB of LDS per CU free. The
# decode depends on no activation: the only thing that orders it is the scratch buffer it writes. So the requests of a
# step are recorded once (the sequence is identical from step to step: frozen weights, fixed layer order, also under
# either checkpointing style) and replayed as a plan: serving request i enqueues the decode of request i+1 on the side
# stream into the other of TWO scratch slots, gated by an event that covers the consumer of request i-1 (the last
# reader of that slot); request i+1 then only makes the main stream wait for that decode's event.
# A misprediction (another call sequence: evaluation, generation prefill) falls back to decoding in stream order, after
# waiting for the mispredicted decode (same slot: write-after-write); nothing is ever read that was not decoded from
# exactly the quant states the caller passed. No prefetch crosses a step boundary (weights may be reloaded there).
# The reference has no counterpart (bitsandbytes' decode is a blocking call on the current stream, utils.py:650-675).


class _HipBackend:
    """Streams / events / the decode launch, as the scheduler sees them (tests substitute a simulator)."""

    def __init__(self, device):
        self.device = device
        self.side = torch.cuda.Stream(device=device)

    def main(self):
        return torch.cuda.current_stream(self.device)

    def record(self, stream):
        e = torch.cuda.Event()
        e.record(stream)
        return e

    def wait(self, stream, event):
        stream.wait_event(event)

    def alloc(self, numel, dtype):
        return torch.empty(numel, dtype=dtype, device=self.device)

    def release(self, buf):
        # a side-stream decode may still be writing the old buffer: nothing may be handed back to the caching
        # allocator (whose blocks are reusable at once by main-stream allocations) before the device is idle
        torch.cuda.synchronize(self.device)

    def decode(self, stream, packed_list, qs_list, buf2d):
        with torch.cuda.stream(stream):
            r = 0
            for pk, q in zip(packed_list, qs_list):
                dequantize_nf4(pk, q, out=buf2d[r:r + q.shape[0]])
                r += q.shape[0]

    def decodable_off_stream(self, qs_list):
        # the cached fp32 absmax must exist already: building it on the side stream would race its first main-stream use
        return all((not q.nested) or getattr(q, "_absmax_f32", None) is not None for q in qs_list)


class _PlanEntry:
    __slots__ = ("key", "packed", "qs")

    def __init__(self, key, packed_list, qs_list):
        self.key = key
        self.packed = [_weakref.ref(p) for p in packed_list]
        self.qs = [_weakref.ref(q) for q in qs_list]


class DecodeAhead:
    """Per-device scheduler. `fetch` returns the stacked row-major decode [W_1; W_2; ...] of a projection group in one of
    two scratch slots; the result is valid until the second next fetch on this device (the caller's GEMM is enqueued
    before the next fetch: every call site launches its consumer right after the decode)."""

    def __init__(self, backend):
        self.be = backend
        self.bufs = [None, None]
        self.count = 0                 # requests served; request n lives in slot n & 1
        self.pending = None            # (key, qs_list, slot, event, buf2d): decode in flight for the NEXT request
        self.plan, self.cur, self.pos, self.in_sync = None, [], 0, False
        self.hits = self.misses = self.prefetches = 0
        self.max_numel = 0

    def reset(self):
        if self.pending is not None:
            self.be.wait(self.be.main(), self.pending[3])
        self.pending = None
        self.plan, self.cur, self.pos, self.in_sync = None, [], 0, False

    def step_begin(self):
        """Top-level model forward: the requests recorded since the last call become the plan of this step."""
        if self.cur:
            self.plan = self.cur
        self.cur, self.pos, self.in_sync = [], 0, self.plan is not None
        if self.pending is not None:                       # never carried across a step boundary
            self.be.wait(self.be.main(), self.pending[3])
            self.pending = None
        if self.plan is not None and self.max_numel:
            # both slots as large as the largest request seen, so that a prefetch never finds its slot too small
            for slot in (0, 1):
                b = self.bufs[slot]
                if b is not None and b.numel() < self.max_numel:
                    self._slot(slot, self.max_numel, b.dtype, grow=True)

    def _slot(self, slot, numel, dtype, grow):
        buf = self.bufs[slot]
        if buf is None or buf.dtype != dtype or buf.numel() < numel:
            if not grow:
                return None
            if buf is not None:
                self.be.release(buf)
            buf = self.be.alloc(numel, dtype)
            self.bufs[slot] = buf
        return buf

    def fetch(self, packed_list, qs_list):
        be = self.be
        first = qs_list[0]
        cols, dtype = first.shape[1], first.dtype
        rows = sum(q.shape[0] for q in qs_list)
        key = tuple(id(q) for q in qs_list)
        slot = self.count & 1
        self.count += 1
        main = be.main()
        pend, self.pending = self.pending, None
        if (pend is not None and pend[2] == slot and pend[0] == key and len(pend[1]) == len(qs_list)
                and all(a is b for a, b in zip(pend[1], qs_list)) and pend[4].dtype == dtype):
            be.wait(main, pend[3])
            buf2d = pend[4]
            self.hits += 1
        else:
            if pend is not None:
                be.wait(main, pend[3])                     # mispredicted decode in flight: order the overwrite behind it
            self.max_numel = max(self.max_numel, rows * cols)
            buf = self._slot(slot, self.max_numel, dtype, grow=True)
            buf2d = buf[:rows * cols].view(rows, cols)
            be.decode(main, packed_list, qs_list, buf2d)
            self.misses += 1
        # ---- record, follow the plan, start the next decode
        self.cur.append(_PlanEntry(key, packed_list, qs_list))
        nxt = None
        if self.in_sync:
            if self.pos < len(self.plan) and self.plan[self.pos].key == key:
                self.pos += 1
                if self.pos < len(self.plan):
                    nxt = self.plan[self.pos]
            else:
                self.in_sync = False
        if nxt is not None:
            self._prefetch(nxt, main)
        views, r = [], 0
        for q in qs_list:
            views.append(buf2d[r:r + q.shape[0]])
            r += q.shape[0]
        return buf2d, views

    def _prefetch(self, ent, main):
        be = self.be
        pk = [r() for r in ent.packed]
        qs = [r() for r in ent.qs]
        if any(x is None for x in pk) or any(x is None for x in qs) or not be.decodable_off_stream(qs):
            return
        cols, dtype = qs[0].shape[1], qs[0].dtype
        rows = sum(q.shape[0] for q in qs)
        slot = self.count & 1
        buf = self._slot(slot, rows * cols, dtype, grow=False)
        if buf is None:                                    # slot too small: that request will miss and grow it
            return
        buf2d = buf[:rows * cols].view(rows, cols)
        # everything enqueued on the main stream so far -- in particular the consumer of the request that used this
        # slot last (two requests back) -- precedes the overwrite
        be.wait(be.side, be.record(main))
        be.decode(be.side, pk, qs, buf2d)
        self.pending = (ent.key, qs, slot, be.record(be.side), buf2d)
        self.prefetches += 1


