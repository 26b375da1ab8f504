"""Full fine-tuning across data-parallel replicas (BASELINE config 3: Llama-3-8B, bf16, DP = 8; SURVEY 8(e) last lines).

The reference hands `full_finetuning=True` to HF's Trainer: torch.nn.Linear under autograd, DDP's bucketed all-reduce of
8.03 B bf16 gradients (16 GB), one optimizer replica per rank (loader.py:487-523 -> FastModel; vision.py:1247-1268,
2174-2209). MI355X-first layout of the same step:

  * every trainable parameter LIVES in a flat per-decoder-layer bucket of the model's dtype (`p.data` is a view), its
    gradient in a twin bucket (`p.grad` is a view): the weight-gradient GEMMs (uamd_gemm_tn_256) write straight into the
    bucket -- no per-parameter AccumulateGrad kernel, no flatten / unflatten copies, q|k|v gradients are one GEMM;
  * buckets complete in backward order (head, layer L-1 .. 0, embedding); a complete bucket is REDUCE-SCATTERED at once
    (SUM over ranks, RCCL over xGMI; 436 MB per Llama-3-8B layer: bandwidth-bound, the full-mesh direct algorithm keeps
    all 7 links busy) on RCCL's stream while the next layer's backward runs;
  * the optimizer is SHARDED (ZeRO-1): a rank keeps fp32 master weights + AdamW moments for 1/N of every bucket only
    (12 B x 8.03 B / 8 = 12 GB per GPU instead of 96 GB), updates its slice with ONE launch per bucket
    (uamd_adamw_shard) and the updated 16-bit slices are ALL-GATHERED back into the parameter buckets, bucket by bucket,
    overlapping the remaining updates;
  * gradients are pre-normalised by the GLOBAL token count (dp.global_num_items), so the reduction is a plain SUM
    (reference semantics: num_items_in_batch, _utils.py:3142-3197);
  * on ONE GPU the same code runs without collectives: 288 GB of HBM hold bf16 weights + gradients (32 GB) and the
    fp32 master + moments (96 GB) of the whole 8B model.
Works unchanged on gloo / CPU (all-reduce + slice instead of reduce-scatter; torch arithmetic instead of the HIP
kernels), which is how tests/test_full_finetune_gloo.py covers the N > 1 path.
"""
import math
import os
import re
import weakref

import torch
import torch.distributed as dist


def _layer_index(name):
    m = re.search(r"(?:^|\.)layers\.(\d+)\.", name)
    return int(m.group(1)) if m else None


def _bucket_key(name, n_layers):
    """Backward order: 0 = head (lm_head, final norm), 1 + (n_layers - 1 - l) = decoder layer l, last = embeddings."""
    li = _layer_index(name)
    if li is not None:
        return 1 + (n_layers - 1 - li)
    return n_layers + 1 if "embed" in name else 0


class FullGradBuckets:
    """Flat parameter + gradient buckets of a fully trainable model and their reduce-scatter exchange."""

    ALIGN = 64                # elements: every shard starts on a 128-byte boundary

    def __init__(self, model, process_group=None, overlap=True):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        dtypes = {p.dtype for _, p in named}
        if len(dtypes) != 1:
            raise TypeError(f"full fine-tuning buckets hold ONE dtype, the model has {sorted(map(str, dtypes))}")
        self.dtype = dtypes.pop()
        self.device = named[0][1].device
        self.group = process_group
        self.overlap = overlap
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self._gloo = dist.is_initialized() and dist.get_backend(process_group) == "gloo"
        # UNSLOTH_AMD_DP_FORCE=1: issue the collectives even in a 1-rank group (exercises RCCL's in-place reduce-scatter /
        # all-gather and the hook ordering on one GPU, tests/test_gpu_dp_rccl.py)
        self._exchange = self.world_size > 1 or (os.environ.get("UNSLOTH_AMD_DP_FORCE", "0") == "1" and dist.is_initialized())
        layers = [l for l in (_layer_index(n) for n, _ in named) if l is not None]
        n_layers = (max(layers) + 1) if layers else 0
        groups = {}
        for n, p in named:
            groups.setdefault(_bucket_key(n, n_layers), []).append((n, p))
        self.buckets = []                      # dicts: names, params, offsets, numel (padded), flat_p, flat_g, shard
        self._where = {}                       # id(p) -> (bucket index, offset)
        quantum = self.ALIGN * self.world_size
        tied = {id(model.get_output_embeddings().weight)} if (
            hasattr(model, "get_output_embeddings") and model.get_output_embeddings() is not None
            and hasattr(model, "get_input_embeddings")
            and model.get_output_embeddings().weight is model.get_input_embeddings().weight) else set()
        with torch.no_grad():
            for key in sorted(groups):
                items = groups[key]
                offs, off = [], 0
                for _, p in items:
                    offs.append(off)
                    off += (p.numel() + 7) // 8 * 8                      # every view 16-byte aligned
                padded = (off + quantum - 1) // quantum * quantum
                flat_p = torch.zeros(padded, dtype=self.dtype, device=self.device)
                flat_g = torch.zeros(padded, dtype=self.dtype, device=self.device)
                # gradient arrivals that complete the bucket: ONE per parameter and backward. A tied embedding has two
                # producers (lm_head's gradient and the embedding lookup's), but autograd sums them before AccumulateGrad:
                # the post-accumulate hook fires once (tests/test_full_finetune_gloo.py counts it)
                b = dict(names=[n for n, _ in items], params=[p for _, p in items], offsets=offs, numel=padded,
                         flat_p=flat_p, flat_g=flat_g, shard=padded // self.world_size, pending=0, handle=None,
                         launched=False, expected=len(items))
                bi = len(self.buckets)
                for (n, p), o in zip(items, offs):
                    k = p.numel()
                    flat_p[o:o + k].copy_(p.data.reshape(-1))
                    p.data = flat_p[o:o + k].view(p.shape)               # the parameter now LIVES in the bucket
                    p.grad = None
                    self._where[id(p)] = (bi, o)
                self.buckets.append(b)
        self.params = [p for b in self.buckets for p in b["params"]]
        self._views = {id(p): self.buckets[bi]["flat_g"][o:o + p.numel()].view(p.shape)
                       for p in self.params for (bi, o) in [self._where[id(p)]]}
        self._written = set()                  # ids whose gradient view holds THIS step's gradient
        self._arrived = set()                  # ids counted towards their bucket in THIS backward
        self._sync = True
        self.timing = False                    # wait() brackets itself with events: exposed_ms() (see dp.LoRAGradArena)
        self._wait_events = []
        self.collectives = 0                   # reduce-scatters issued so far (telemetry: tests, bench)
        self._hooks = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        # direct sinks (kernels/utils.GRAD_SINKS): the weight-gradient GEMM / the norm dW kernel / the chunked lm_head
        # gradient write into the bucket themselves. Tied embeddings keep the autograd path (two producers).
        self._direct = set()
        if self.device.type == "cuda":
            from .kernels.utils import GRAD_SINKS
            for k in [k for k, ref in GRAD_SINKS.items() if ref() is None]:
                del GRAD_SINKS[k]
            me = weakref.ref(self)
            for p in self.params:
                if id(p) not in tied:
                    GRAD_SINKS[id(p)] = me
                    self._direct.add(id(p))
        me_ = weakref.ref(self)
        for p in self.params:
            p._uamd_arena = me_

    # ---- sink interface (kernels/fast_dense.weight_grads, rms_layernorm._weight_grad, cross_entropy_loss) -------------
    def grad_view(self, p):
        return self._views[id(p)]

    def first_write(self, p):
        """True when the view holds nothing of this step yet: the producer OVERWRITES it (no zero fill, no read of 16 GB
        of zeros); False: it accumulates (gradient accumulation over micro-batches, a second producer)."""
        return id(p) not in self._written

    def ready(self, p):
        """The kernel that wrote / added into grad_view(p) has been enqueued."""
        self._written.add(id(p))
        p.grad = self._views[id(p)]
        self._count(p)

    # ---- autograd path (embeddings; everything on the CPU) ------------------------------------------------------------
    def _hook(self, p):
        v = self._views[id(p)]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            if id(p) in self._written:
                v.add_(p.grad)
            else:
                v.copy_(p.grad)
            p.grad = v
        self._written.add(id(p))
        self._count(p)

    def _count(self, p):
        # one arrival per parameter and backward: a sink reports through ready() AND torch still runs the post-accumulate hook of
        # a parameter whose Function returned None (see dp.LoRAGradArena._hook: every bucket was exchanged twice, the first time
        # half complete)
        if id(p) in self._arrived:
            return
        self._arrived.add(id(p))
        bi, _ = self._where[id(p)]
        b = self.buckets[bi]
        b["pending"] += 1
        if b["pending"] == b["expected"]:
            b["pending"] = 0
            if self._sync and self.overlap and self._exchange and not b["launched"]:
                self._launch(bi)

    # ---- exchange ------------------------------------------------------------------------------------------------------
    def grad_shard(self, bi):
        b = self.buckets[bi]
        return b["flat_g"][self.rank * b["shard"]:(self.rank + 1) * b["shard"]]

    def param_shard(self, bi):
        b = self.buckets[bi]
        return b["flat_p"][self.rank * b["shard"]:(self.rank + 1) * b["shard"]]

    def _launch(self, bi):
        b = self.buckets[bi]
        if self._gloo:                         # gloo has no reduce-scatter: all-reduce, every rank reads its own slice
            b["handle"] = dist.all_reduce(b["flat_g"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:                                  # in place: the rank's slice of the bucket receives the sum
            b["handle"] = dist.reduce_scatter_tensor(self.grad_shard(bi), b["flat_g"], op=dist.ReduceOp.SUM,
                                                     group=self.group, async_op=True)
        b["launched"] = True
        self.collectives += 1

    def finish(self):
        """After backward, before the optimizer: parameters that received no gradient this step count as zero; buckets
        without a collective in flight get one (every rank must exchange the SAME buckets); nothing is waited for here --
        the optimizer waits bucket by bucket."""
        for p in self.params:
            if id(p) not in self._written:
                self._views[id(p)].zero_()
                p.grad = self._views[id(p)]
        if self._exchange and self._sync:
            for bi, b in enumerate(self.buckets):
                if not b["launched"]:
                    self._launch(bi)
        self._reset_arrivals()

    def _reset_arrivals(self):
        """Arrival counts never leak from one backward into the next: a bucket some parameter of which received no gradient
        stays below `expected`, and the leftover would complete it in the MIDDLE of the next micro-batch's backward -- a
        reduce-scatter of half-accumulated gradients. Called where a backward is known to be over: finish(), the exit of a
        no_sync() block (one micro-batch per block, the accelerate / HF Trainer pattern), zero_grad()."""
        for b in self.buckets:
            b["pending"] = 0
        self._arrived.clear()

    def wait(self, bi):
        b = self.buckets[bi]
        if b["handle"] is not None:
            timed = self.timing and self.device.type == "cuda"
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            b["handle"].wait()
            if timed:
                e1.record()
                self._wait_events.append((e0, e1))
            b["handle"] = None
        b["launched"] = False

    def exposed_ms(self, reset=True):
        """Time the compute stream spent blocked on reduce-scatters since the last reset (timing = True; after a synchronize)."""
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events)
        if reset:
            self._wait_events = []
        return ms

    def gather_params(self, bi, async_op=True):
        """All-gather the updated parameter slices of bucket `bi` back into its flat parameter buffer (in place)."""
        if not self._exchange:
            return None
        b = self.buckets[bi]
        return dist.all_gather_into_tensor(b["flat_p"], self.param_shard(bi), group=self.group, async_op=async_op)

    def zero_grad(self):
        """Nothing is filled: the next step's first producer of every gradient overwrites its view."""
        self._written.clear()
        for p in self.params:
            p.grad = None
        self._reset_arrivals()

    def no_sync(self):
        """Gradient accumulation: no exchange for the backward(s) inside the block -- wrap ONE micro-batch per block."""
        from contextlib import contextmanager

        @contextmanager
        def ctx():
            old, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = old
                self._reset_arrivals()
        return ctx()

    def close(self):
        if self.device.type == "cuda":
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

    def describe(self):
        return dict(buckets=len(self.buckets), bytes=[b["numel"] * self.buckets[0]["flat_p"].element_size() for b in self.buckets],
                    params=len(self.params), world_size=self.world_size, dtype=str(self.dtype))


class ShardedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW's arithmetic on fp32 master weights, sharded over the data-parallel group (see module docstring).
    One parameter group; LR schedulers work through the usual Optimizer interface (they scale the group's `lr`); the state of
    the flat shards (step count, fp32 masters, both moments) is saved and loaded under the extra `uamd_sharded` key of
    state_dict(), per rank. Weight decay follows HF Trainer's rule for the reference's full_finetuning path
    (`get_decay_parameter_names`: no decay on biases and on LayerNorm / RMSNorm weights): one-dimensional parameters are
    updated with decay 0 -- they sit at the end of every per-layer bucket, so a bucket is two launches instead of one."""

    def __init__(self, model_or_buckets, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, process_group=None):
        self.buckets = model_or_buckets if isinstance(model_or_buckets, FullGradBuckets) else \
            FullGradBuckets(model_or_buckets, process_group=process_group)
        super().__init__(list(self.buckets.params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        B = self.buckets
        self.master, self.exp_avg, self.exp_avg_sq = [], [], []
        for bi, b in enumerate(B.buckets):
            self.master.append(B.param_shard(bi).to(torch.float32))            # (a copy: the shard changes dtype)
            self.exp_avg.append(torch.zeros(b["shard"], dtype=torch.float32, device=B.device))
            self.exp_avg_sq.append(torch.zeros(b["shard"], dtype=torch.float32, device=B.device))
        # [start, end, decays) runs of this rank's shard of every bucket, in shard-local elements
        self._runs = []
        for bi, b in enumerate(B.buckets):
            lo, hi = B.rank * b["shard"], (B.rank + 1) * b["shard"]
            marks = []                                   # (offset in the bucket, decays) per parameter, in layout order
            for p, o in zip(b["params"], b["offsets"]):
                marks.append((o, p.dim() > 1))
            runs = []
            for i, (o, dec) in enumerate(marks):
                end = marks[i + 1][0] if i + 1 < len(marks) else b["numel"]
                a, e = max(o, lo), min(end, hi)
                if a >= e:
                    continue
                if runs and runs[-1][2] == dec and runs[-1][1] == a - lo:
                    runs[-1][1] = e - lo
                else:
                    runs.append([a - lo, e - lo, dec])
            self._runs.append(runs)
        self._t = 0

    @property
    def arena(self):                      # trainer.unsloth_train adopts `optimizer.arena` as the exchange object
        return self.buckets

    def grad_norm(self):
        """Global L2 norm of the (reduced) gradient: each rank's shard, summed over ranks. Waits for the exchange."""
        B = self.buckets
        sq = torch.zeros((), dtype=torch.float32, device=B.device)
        for bi in range(len(B.buckets)):
            B.wait(bi)
            sq += B.grad_shard(bi).float().pow(2).sum()
        if B.world_size > 1:
            dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=B.group)
        return sq.sqrt()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grp = self.param_groups[0]
        b1, b2 = grp["betas"]
        self._t += 1
        bc1 = 1.0 - b1 ** self._t
        bc2_sqrt = math.sqrt(1.0 - b2 ** self._t)
        B = self.buckets
        handles = []
        for bi in range(len(B.buckets)):                 # backward order: the first bucket's exchange finished first
            B.wait(bi)
            g16, p16 = B.grad_shard(bi), B.param_shard(bi)
            p32, m, v = self.master[bi], self.exp_avg[bi], self.exp_avg_sq[bi]
            for a, e, decays in self._runs[bi]:
                wd = float(grp["weight_decay"]) if decays else 0.0
                if p32.is_cuda:
                    from . import _lib
                    es = g16.element_size()
                    with _lib.device_ctx(p32):
                        rc = _lib.lib().uamd_adamw_shard(
                            p32.data_ptr() + 4 * a, g16.data_ptr() + es * a, p16.data_ptr() + es * a, m.data_ptr() + 4 * a,
                            v.data_ptr() + 4 * a, e - a, float(grp["lr"]), float(b1), float(b2), float(grp["eps"]), wd, bc1,
                            bc2_sqrt, float(grad_scale), _lib.dtype_code(g16.dtype), _lib.stream_of(p32))
                    _lib.check(rc, "uamd_adamw_shard")
                else:                                    # host arithmetic of the gloo tests: the same formula in torch
                    g = g16[a:e].to(torch.float32) * grad_scale
                    p32[a:e].mul_(1.0 - grp["lr"] * wd)
                    m[a:e].mul_(b1).add_(g, alpha=1.0 - b1)
                    v[a:e].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                    p32[a:e].addcdiv_(m[a:e], v[a:e].sqrt() / bc2_sqrt + grp["eps"], value=-grp["lr"] / bc1)
                    p16[a:e].copy_(p32[a:e])
            h = B.gather_params(bi)
            if h is not None:
                handles.append(h)
        for h in handles:
            h.wait()
        return loss

    def zero_grad(self, set_to_none=True):
        self.buckets.zero_grad()

    def state_dict(self):
        sd = super().state_dict()
        sd["uamd_sharded"] = dict(step=self._t, rank=self.buckets.rank, world_size=self.buckets.world_size,
                                  master=[t.clone() for t in self.master], exp_avg=[t.clone() for t in self.exp_avg],
                                  exp_avg_sq=[t.clone() for t in self.exp_avg_sq])
        return sd

    def load_state_dict(self, state_dict):
        sh = state_dict.get("uamd_sharded")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "uamd_sharded"})
        if sh is None:
            return
        if sh["world_size"] != self.buckets.world_size or sh["rank"] != self.buckets.rank:
            raise ValueError("ShardedAdamW state was saved by another rank / world size: every rank loads its own shard")
        self._t = int(sh["step"])
        with torch.no_grad():
            for dst, src in ((self.master, sh["master"]), (self.exp_avg, sh["exp_avg"]), (self.exp_avg_sq, sh["exp_avg_sq"])):
                for d, s_ in zip(dst, src):
                    d.copy_(s_)
            for bi in range(len(self.buckets.buckets)):
                self.buckets.param_shard(bi).copy_(self.master[bi])
                self.buckets.gather_params(bi, async_op=False)


def full_finetune_step(model, batch, optimizer, num_items=None):
    """One optimizer step on one micro-batch (the full fine-tuning counterpart of trainer.training_step)."""
    from .dp import global_num_items
    if num_items is None:
        num_items = global_num_items(batch["labels"], optimizer.buckets.group)
    loss = model(**batch, num_items_in_batch=num_items).loss
    loss.backward()
    optimizer.buckets.finish()
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach()
