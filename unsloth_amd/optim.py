"""FlatAdamW: torch.optim.AdamW's arithmetic over ONE flat fp32 arena -- one HIP launch per optimizer step.

The reference trains through HF's Trainer with a torch / bitsandbytes optimizer (unsloth/trainer.py:445-623); the
benchmark's step is forward + backward + AdamW on the LoRA factors. torch's fused AdamW needs 52 multi_tensor_apply
launches (2.5 ms) for the 448 factors of Llama-3-8B r=16 -- 168 MB of parameters, 0.2 ms of HBM time. MI355X-first
layout instead: parameters, gradients (dp.LoRAGradArena, which the fused LoRA-gradient kernel ADDS into), exp_avg and
exp_avg_sq are four flat fp32 buffers with the same offsets, `p.data` / `p.grad` / the optimizer state are views into
them, and `uamd_adamw_flat` (csrc/adamw.hip) walks them once -- zeroing the gradient arena in the same pass, which is
the `zero_grad` of the next step.

Interface: a torch.optim.Optimizer (param_groups / state / state_dict / LR schedulers / step hooks work as usual) with
one parameter group. `step(grad_scale=...)` takes the clipping factor; parameters whose gradient is None are skipped
like torch does (all parameters share one step counter: in LoRA training every factor gets a gradient every step). No CPU fallback: the step raises without the HIP library.
"""
import math

import torch

from . import _lib
from .dp import LoRAGradArena


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, arena=None):
        """`arena`: the dp.LoRAGradArena that already owns the gradients (data-parallel runs); else one is created
        (single rank: no collective is ever issued)."""
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid AdamW hyper-parameters")
        if arena is None:
            # two live arenas over the same parameters would both hook gradient accumulation and fight over p.grad and
            # the fused-gradient sinks (the second one's copies racing the first one's collectives): refuse
            for p in model.parameters():
                owner = getattr(p, "_uamd_arena", None)
                if p.requires_grad and owner is not None and owner() is not None:
                    raise RuntimeError("FlatAdamW: these parameters already belong to a live dp.LoRAGradArena; pass it "
                                       "as `arena=` (trainer.make_optimizer(model, arena=arena))")
        self.arena = arena if arena is not None else LoRAGradArena(model)
        self._owns_arena = arena is None
        params = list(self.arena.params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        g = self.arena.arena
        n = g.numel()
        self.flat_p = torch.empty(n, dtype=torch.float32, device=g.device)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=g.device)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=g.device)
        self._views = []                     # (param, offset, numel, grad view)
        self._step_t = torch.zeros((), dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                gv = self.arena._views[id(p)]
                assert gv.data_ptr() == g.data_ptr() + 4 * off, "arena order changed under the optimizer"
                self.flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)          # the parameter now LIVES in the flat buffer
                self.state[p] = dict(step=self._step_t,          # ONE shared host scalar: one increment per step, not 448
                                     exp_avg=self.flat_m[off:off + k].view(p.shape),
                                     exp_avg_sq=self.flat_v[off:off + k].view(p.shape))
                self._views.append((p, off, k, gv))
                off += k
        self._t = 0
        self._writes_seen = self.arena.writes    # arena.writes at the moment the arena was last known to be all zeros

    # ------------------------------------------------------------------------------------------
    def _runs(self):
        """Contiguous [start, end) element ranges whose parameters have a gradient; every gradient is (moved) in the
        arena first. One range = the whole arena in a normal LoRA step."""
        runs, cur = [], None
        base = self.flat_p.data_ptr()
        for p, off, k, gv in self._views:
            if p.data_ptr() != base + 4 * off:
                # someone re-pointed p.data (module.to(...), a checkpoint loader that assigns .data): adopt the new
                # values and bring the parameter home again -- stepping a detached copy would silently train nothing
                self.flat_p[off:off + k].copy_(p.data.reshape(-1).to(self.flat_p.dtype))
                p.data = self.flat_p[off:off + k].view(p.shape)
            gr = p.grad
            if gr is None:
                cur = None
                continue
            if gr.data_ptr() != gv.data_ptr():
                gv.copy_(gr)                 # autograd built a fresh tensor (the view had been dropped)
                p.grad = gv
            if cur is not None and cur[1] == off:
                cur[1] = off + k
            else:
                cur = [off, off + k]
                runs.append(cur)
        return runs

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grp = self.param_groups[0]
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatAdamW: one parameter group (use torch.optim.AdamW for per-group settings)")
        b1, b2 = grp["betas"]
        self._t += 1
        t = self._t
        bc1 = 1.0 - b1 ** t
        bc2_sqrt = math.sqrt(1.0 - b2 ** t)
        runs = self._runs()
        _lib.require_gpu(self.flat_p)
        g = self.arena.arena
        L = _lib.lib()
        with _lib.device_ctx(self.flat_p):
            for s, e in runs:
                # (a run that does not start on a 16-byte boundary cannot happen for LoRA factors: every numel is a
                # multiple of 4; the C side checks)
                rc = L.uamd_adamw_flat(self.flat_p.data_ptr() + 4 * s, g.data_ptr() + 4 * s, self.flat_m.data_ptr() + 4 * s,
                                       self.flat_v.data_ptr() + 4 * s, e - s, float(grp["lr"]), float(b1), float(b2),
                                       float(grp["eps"]), float(grp["weight_decay"]), bc1, bc2_sqrt, float(grad_scale), 1,
                                       _lib.stream_of(self.flat_p))
                _lib.check(rc, "uamd_adamw_flat")
        self._step_t += 1                    # (shared by every parameter's state entry)
        # every range that had a gradient is zero again; ranges without one were never written
        self._writes_seen = self.arena.writes
        return loss

    def zero_grad(self, set_to_none=True):
        """The step already zeroed the arena in its own pass; the gradient views stay attached (the fused LoRA-gradient
        kernel adds into them). Called without a step in between (gradients thrown away): one fill."""
        if self.arena.writes != self._writes_seen:
            self.arena.arena.zero_()
            self._writes_seen = self.arena.writes
        self.arena.reset_arrivals()                  # (arrival bookkeeping of the exchange: a new accumulation starts)
        for p, _, _, gv in self._views:
            p.grad = gv

    def grad_norm(self):
        return self.arena.arena.norm()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch replaced the state tensors by copies: move them back into the flat buffers and re-attach the views.
        # The step counter becomes the LOADED one (loading an earlier checkpoint into an optimizer that has already
        # stepped must restart the bias correction there); a parameter the checkpoint has no state for (saved before the
        # first step) starts from zero moments. All parameters share one step counter -- a parameter that received no
        # gradient on some steps is bias-corrected with the run's step count, not its own (torch counts per parameter;
        # in LoRA training every factor gets a gradient every step, so the two agree).
        loaded_t = None
        with torch.no_grad():
            for p, off, k, _ in self._views:
                st = self.state.get(p, None)
                if not st or "exp_avg" not in st:
                    self.flat_m[off:off + k].zero_()
                    self.flat_v[off:off + k].zero_()
                    st = self.state[p] = {}
                else:
                    self.flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    self.flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    if "step" in st:
                        t = int(st["step"])
                        loaded_t = t if loaded_t is None else max(loaded_t, t)
                st["exp_avg"] = self.flat_m[off:off + k].view(p.shape)
                st["exp_avg_sq"] = self.flat_v[off:off + k].view(p.shape)
                st["step"] = self._step_t
            self._t = 0 if loaded_t is None else loaded_t
            self._step_t.fill_(float(self._t))

    def close(self):
        if self._owns_arena:
            self.arena.close()
