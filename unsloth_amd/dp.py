"""Data-parallel glue for QLoRA replicas.

The reference has no collective call sites: PyTorch DDP (via accelerate) all-reduces whatever
requires grad, and unsloth only pins one replica per device (`prepare_device_map`,
models/loader_utils.py:91-106), excludes rotary buffers from DDP (:849-865) and suppresses
nn.DataParallel (models/_utils.py:187-249). The frozen NF4 base is replicated; the ONLY exchange per
optimizer step is the sum of the LoRA gradients (41,943,040 fp32 = 167.8 MB for Llama-3-8B r=16).

MI355X-first version of that exchange:
  * all trainable grads live in ONE contiguous fp32 arena (p.grad are views), so a bucket is a slice,
    not a flatten/unflatten copy;
  * buckets follow the decoder-layer order (>= bucket_bytes each); a bucket is all-reduced (SUM, RCCL
    over xGMI) from the post-accumulate hook of its LAST gradient, i.e. while the remaining layers'
    backward is still running (async collective on RCCL's stream);
  * the loss is normalised by the GLOBAL token count on every rank, so the reduction is a plain SUM
    with no 1/world_size (reference semantics: num_items_in_batch, _utils.py:3142-3197);
  * deterministic: fixed bucket order, fp32 sum.
Works unchanged on gloo/CPU (tests/test_dp_gloo.py, world_size 2).
"""
import os
import re
from contextlib import contextmanager

import torch
import torch.distributed as dist


def _layer_index(name):
    m = re.search(r"\.layers\.(\d+)\.", name)
    return int(m.group(1)) if m else -1


class LoRAGradArena:
    def __init__(self, model, process_group=None, bucket_bytes=None, overlap=True, direct=True):
        if bucket_bytes is None:
            # a bucket closes at the first decoder-layer boundary past this size: Llama-3-8B r=16 factors are 5.24 MB per
            # layer, so 16 MB means FOUR layers = 21 MB per bucket; the LAST exchange is the only one nothing can overlap
            # (runnable 0.4 ms before the optimizer kernel), so the first decoder layer is a bucket of its own
            # (below): 9 buckets = 7 x 21 MB, layers 3..1 (15.7 MB), layer 0 (5.2 MB) -- all far above the size where an xGMI
            # ring is latency-bound
            bucket_bytes = int(float(os.environ.get("UNSLOTH_AMD_DP_BUCKET_MB", "16")) * (1 << 20))
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        # backward produces the LAST layer first: buckets are laid out in that order
        named.sort(key=lambda np_: -_layer_index(np_[0]))
        self.params = [p for _, p in named]
        self.names = [n for n, _ in named]
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.arena = torch.zeros(total, dtype=torch.float32, device=dev)
        self.group = process_group
        self.overlap = overlap
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # UNSLOTH_AMD_DP_FORCE=1: issue the collectives even in a 1-rank group (exercises RCCL + hooks on one GPU)
        self._force = os.environ.get("UNSLOTH_AMD_DP_FORCE", "0") == "1" and dist.is_initialized()
        self.buckets = []          # (start, end, n_params)
        self._bucket_of = {}
        off, b_start, b_count = 0, 0, 0
        last_layer = None
        # the gradients that arrive LAST (the first decoder layer's: backward ends there) get a bucket of their own: its
        # all-reduce is the one exchange nothing can overlap -- it becomes runnable when the backward is over -- so it should be
        # as small as a bucket gets (one layer: 5.2 MB for Llama-3-8B r=16 instead of the 21 MB of a four-layer bucket)
        final_layer = _layer_index(named[-1][0])
        for n, p in named:
            layer = _layer_index(n)
            if b_count and layer != last_layer and ((off - b_start) * 4 >= bucket_bytes or layer == final_layer):
                self.buckets.append([b_start, off, b_count])
                b_start, b_count = off, 0
            if p.dtype != torch.float32:
                raise TypeError(f"{n}: LoRA parameters are expected in fp32 (SURVEY 9.10), got {p.dtype}")
            p.grad = self.arena[off:off + p.numel()].view_as(p)
            self._bucket_of[id(p)] = len(self.buckets)
            off += p.numel()
            b_count += 1
            last_layer = layer
        self.buckets.append([b_start, off, b_count])
        self._pending = [0] * len(self.buckets)
        self._arrived = set()                              # ids of the parameters whose gradient arrived in this backward
        self._launched = [False] * len(self.buckets)      # per step: which buckets already have a collective in flight
        self.collectives = 0                               # all-reduces issued so far (telemetry: bench.py, tests)
        self.writes = 0                                    # gradients accumulated so far (optim.FlatAdamW: "is the arena dirty?")
        self._handles = []
        self._sync = True
        # timing = True: finish() brackets its waits with events on the compute stream -- the time the step is BLOCKED on
        # the exchange (what the overlap did not hide); read with exposed_ms() after a synchronize (bench.py `dp_exposed_ms`)
        self.timing = False
        self._wait_events = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        self._views = {id(p): p.grad for p in self.params}
        import weakref as _weakref
        me_ = _weakref.ref(self)
        for p in self.params:
            p._uamd_arena = me_              # "this parameter's gradient lives in a live arena" (optim.FlatAdamW checks it)
        # the fused LoRA-gradient kernel adds straight into the arena (kernels/utils.py GRAD_SINKS): no
        # AccumulateGrad kernel per parameter; .ready() does the bucket bookkeeping the autograd hook would do
        if direct and self.arena.is_cuda:
            import weakref
            from .kernels.utils import GRAD_SINKS
            for k in [k for k, ref in GRAD_SINKS.items() if ref() is None]:      # arenas that died without close()
                del GRAD_SINKS[k]
            me = weakref.ref(self)        # weak: the registry must not keep an arena (and through it a model) alive
            for p in self.params:
                GRAD_SINKS[id(p)] = me

    def grad_view(self, p):
        """Arena slice the fused gradient kernel ADDS into. If the optimizer dropped the gradient
        (`zero_grad(set_to_none=True)`, PyTorch's default) the slice still holds the previous step's reduced
        values: zero it before re-attaching, so the add starts from nothing."""
        v = self._views[id(p)]
        if p.grad is None:
            v.zero_()
            p.grad = v
        elif p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)                              # a gradient accumulated outside the arena so far
            p.grad = v
        return v

    def ready(self, p):
        self._hook(p)

    def close(self):
        from .kernels.utils import GRAD_SINKS
        for p in self.params:
            ref = GRAD_SINKS.get(id(p))
            if ref is not None and ref() is self:
                del GRAD_SINKS[id(p)]
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            ref = getattr(p, "_uamd_arena", None)
            if ref is not None and ref() is self:
                del p._uamd_arena

    # ------------------------------------------------------------------------------------------
    def _hook(self, p):
        v = self._views[id(p)]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            # autograd's AccumulateGrad built a fresh p.grad (the view had been dropped by
            # zero_grad(set_to_none=True)): move it into the arena, which is what gets reduced and stepped on
            v.copy_(p.grad)
            p.grad = v
        # One arrival per parameter and backward, whoever reports it. With the direct sinks (kernels/utils.py GRAD_SINKS) the
        # fused gradient kernel reports through ready() and hands autograd None -- and torch (2.10) STILL runs the parameter's
        # post-accumulate hook: counted twice, every bucket "completed" when half its gradients were in and was all-reduced
        # twice per step, the first time over partial sums that the second reduction then summed over the ranks AGAIN
        # (found in round 4 on the rocprofv3 trace: 16 collectives for 8 buckets; invisible at one rank, and the CPU tests
        # have no sinks). tests/_dp_worker.py now checks, at one rank too, that a bucket is launched once and complete.
        self.writes += 1                  # (dirtiness of the arena for optim.FlatAdamW.zero_grad: EVERY report counts -- a second
        if id(p) in self._arrived:        #  micro-batch writes too; round 4's first dedup skipped it and an accumulation that
            return                        #  followed a thrown-away backward started from the old sums)
        self._arrived.add(id(p))
        b = self._bucket_of[id(p)]
        self._pending[b] += 1
        if self._pending[b] == self.buckets[b][2]:
            self._pending[b] = 0
            if self._sync and self.overlap and (self.world_size > 1 or self._force) and not self._launched[b]:
                self._launch(b)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        h = dist.all_reduce(self.arena[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append(h)
        self._launched[b] = True
        self.collectives += 1

    def finish(self):
        """Call after backward, before optimizer.step(): launches every bucket that has no collective in flight
        yet -- all of them without overlap, and with overlap the ones whose count never filled because some
        trainable parameter received no gradient this step (every rank must still reduce the SAME buckets, or
        the replicas diverge) -- then waits."""
        if (self.world_size > 1 or self._force) and self._sync:
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launch(b)
            timed = self.timing and self.arena.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for h in self._handles:
                h.wait()
            if timed:
                e1.record()
                self._wait_events.append((e0, e1))
        self._handles = []
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._arrived.clear()

    def exposed_ms(self, reset=True):
        """Sum over the finish() calls since the last reset of the time the compute stream spent waiting for collectives
        (timing = True; call after torch.cuda.synchronize())."""
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events)
        if reset:
            self._wait_events = []
        return ms

    @contextmanager
    def no_sync(self):
        """Gradient accumulation: skip the exchange on all but the last micro-step (one micro-batch per block). Arrival
        counts are dropped at the exit: a bucket that stayed incomplete (some factor received no gradient) must not be
        completed -- and all-reduced half-accumulated -- by the first arrivals of the next micro-batch."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old
            self._pending = [0] * len(self.buckets)
            self._arrived.clear()

    def reset_arrivals(self):
        """A new accumulation starts (the gradients so far were thrown away or stepped on): nobody has arrived yet."""
        self._arrived.clear()
        self._pending = [0] * len(self.buckets)

    def zero_grad(self):
        """Keeps the views: the arena is zeroed in one memset instead of N small ones."""
        self.reset_arrivals()
        self.arena.zero_()
        for p in self.params:
            v = self._views[id(p)]
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def grad_norm(self):
        return self.arena.norm()

    @property
    def nbytes(self):
        return self.arena.numel() * 4

    def describe(self):
        return dict(params=len(self.params), numel=self.arena.numel(), bytes=self.nbytes,
                    buckets=[(e - s) * 4 for s, e, _ in self.buckets], world_size=self.world_size)


def global_num_items(labels, group=None):
    """Non-ignored target count over ALL ranks (what every rank divides its loss sum by)."""
    shift = labels[..., 1:]
    n = torch.count_nonzero(shift != -100).to(torch.int64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return n


_ROTARY_INV_FREQ_BUFFER_NAMES = ("inv_freq", "short_inv_freq", "long_inv_freq", "original_inv_freq")   # the last: transformers 5.x


def exclude_rope_inv_freq_from_ddp(model):
    """loader_utils.py:849-865: a user who wraps the model in torch's DistributedDataParallel anyway (instead of the
    LoRAGradArena exchange above) must not have the HF rotary modules' `inv_freq` buffers broadcast -- they may sit on
    the CPU (transformers v5 leaves non-persistent buffers where meta-init put them) and the product never
    reads them (RopeTables rebuilds cos / sin from the config). Adds their fully qualified names to the list DDP reads
    (`_ddp_params_and_buffers_to_ignore`); re-run after PEFT wrapping, the names change. Returns the model."""
    ignored = list(getattr(model, "_ddp_params_and_buffers_to_ignore", None) or [])
    for module_name, module in model.named_modules():
        for buffer_name, _ in module.named_buffers(recurse=False):
            if buffer_name in _ROTARY_INV_FREQ_BUFFER_NAMES:
                fqn = f"{module_name}.{buffer_name}" if module_name else buffer_name
                if fqn not in ignored:
                    ignored.append(fqn)
    if ignored:
        model._ddp_params_and_buffers_to_ignore = ignored
    return model
