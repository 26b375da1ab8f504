"""Trainer surface; mirror of unsloth/trainer.py.

`UnslothTrainer(SFTTrainer)` / `UnslothTrainingArguments` (trainer.py:445-623) subclass TRL when it is
installed (it is not in this image: then they raise with an explanation), and `unsloth_train`
(trainer.py:49-57) is provided as a minimal self-contained loop over pre-tokenised batches -- forward
through the fused path, backward, LoRA-grad exchange (unsloth_amd/dp.py), optimizer step -- which is
also what bench.py times. Loss normalisation follows the reference's contract: sum of
token losses / global non-ignored token count, passed as `num_items_in_batch`.
"""
import os
import time

import torch

from .dp import LoRAGradArena, global_num_items

try:
    from trl import SFTConfig as _SFTConfig, SFTTrainer as _SFTTrainer
    HAS_TRL = True
except Exception:
    HAS_TRL = False


if HAS_TRL:
    class UnslothTrainingArguments(_SFTConfig):
        def __init__(self, embedding_learning_rate=None, q_galore_config=None, *args, **kwargs):
            self.embedding_learning_rate = embedding_learning_rate
            self.q_galore_config = q_galore_config
            super().__init__(*args, **kwargs)

    class UnslothTrainer(_SFTTrainer):
        """trainer.py:502-623: embedding-LR param groups; everything else is TRL/HF."""

        def create_optimizer(self):
            lr = getattr(self.args, "embedding_learning_rate", None)
            if lr is None or self.optimizer is not None:
                return super().create_optimizer()
            cls, kw = self.get_optimizer_cls_and_kwargs(self.args)
            emb, rest = [], []
            for n, p in self.model.named_parameters():
                if p.requires_grad:
                    (emb if n.endswith("modules_to_save.default.weight") else rest).append(p)
            groups = [dict(params=rest, lr=kw.get("lr", self.args.learning_rate)), dict(params=emb, lr=lr)]
            kw.pop("lr", None)
            self.optimizer = cls(groups, **kw)
            return self.optimizer
else:
    class _NeedsTRL:
        def __init__(self, *a, **k):
            raise ImportError("UnslothTrainer / UnslothTrainingArguments subclass trl.SFTTrainer / SFTConfig "
                              "(unsloth/trainer.py:445-623); `trl` is not installed in this environment. "
                              "Use unsloth_amd.trainer.unsloth_train for pre-tokenised batches.")

    UnslothTrainer = UnslothTrainingArguments = _NeedsTRL


def _config_only_names(config_class):
    """Names a TRL `XConfig` accepts that plain `transformers.TrainingArguments` does not: since TRL 0.13 they can
    no longer be passed to the Trainer itself."""
    import inspect
    from transformers import TrainingArguments
    return set(inspect.signature(config_class).parameters) - set(inspect.signature(TrainingArguments).parameters)


def _backwards_compatible_trainer(trainer_class, config_class):
    """`__init__` wrapper of trainer.py:714-783: scripts written for TRL < 0.13 keep working --
    `tokenizer=` becomes `processing_class=`, and keyword arguments that moved from `XTrainer(...)` to `XConfig(...)`
    (max_seq_length, dataset_text_field, packing, ...) are set on the config object passed as `args`."""
    import functools
    import inspect
    original_init = trainer_class.__init__
    trainer_params = set(inspect.signature(original_init).parameters)

    @functools.wraps(original_init)
    def new_init(self, *args, **kwargs):
        if "processing_class" in trainer_params and "tokenizer" in kwargs and "tokenizer" not in trainer_params:
            kwargs["processing_class"] = kwargs.pop("tokenizer")
        config = kwargs.get("args")
        if config is not None:
            moved = _config_only_names(config_class)
            for key in [k for k in kwargs if k not in trainer_params]:
                if key in moved or hasattr(config, key):
                    setattr(config, key, kwargs.pop(key))
        original_init(self, *args, **kwargs)
    new_init._unsloth_backwards_compatible = True
    return new_init


def _patch_trl_trainer():
    """trainer.py:988-1021: make every `trl.XTrainer` whose `trl.XConfig` exists accept the pre-0.13 calling
    convention. Returns the list of patched trainer names ([] when TRL is absent, too old, or already patched)."""
    try:
        import trl
        import trl.trainer
    except Exception:
        return []
    if hasattr(trl, "__UNSLOTH_BACKWARDS_COMPATIBLE__"):
        return []
    try:
        from packaging.version import Version
        if Version(getattr(trl, "__version__", "0")) <= Version("0.11.0"):
            return []
    except Exception:
        pass
    names = dir(trl.trainer)
    trainers = {n[: -len("Trainer")] for n in names if n.endswith("Trainer")}
    configs = {n[: -len("Config")] for n in names if n.endswith("Config")}
    done = []
    for x in sorted(trainers & configs):
        try:
            tc, cc = getattr(trl, x + "Trainer", None) or getattr(trl.trainer, x + "Trainer"), \
                getattr(trl, x + "Config", None) or getattr(trl.trainer, x + "Config")
            tc.__init__ = _backwards_compatible_trainer(tc, cc)
            done.append(x)
        except Exception:
            continue
    trl.__UNSLOTH_BACKWARDS_COMPATIBLE__ = True
    return done


FLAT_ADAMW = True              # one-launch AdamW over flat arenas (optim.FlatAdamW); False = torch's fused AdamW


def make_optimizer(model, lr=None, weight_decay=0.01, betas=(0.9, 0.999), arena=None, flat=None):
    """AdamW on the trainable (LoRA) parameters. On the GPU with fp32 parameters: optim.FlatAdamW -- parameters,
    gradients and moments in flat arenas, ONE launch per step (`arena`: the dp.LoRAGradArena of a data-parallel run, else
    the optimizer creates its own). `flat=False` (or UNSLOTH_AMD_FLAT_ADAMW=0) keeps torch's fused AdamW."""
    params = [p for p in model.parameters() if p.requires_grad]
    base = model.get_base_model() if hasattr(model, "get_base_model") else model
    if lr is None:             # LoRA factors: 2e-4 (the reference notebooks' rate); every weight of the model: 2e-5
        lr = 2e-5 if getattr(base, "_unsloth_full_finetuning", False) else 2e-4
    if getattr(base, "_unsloth_full_finetuning", False) and params and all(p.is_cuda for p in params):
        # full fine-tuning: flat per-layer buckets + AdamW sharded over the data-parallel group (full_finetune.py)
        from .full_finetune import FullGradBuckets, ShardedAdamW
        return ShardedAdamW(arena if isinstance(arena, FullGradBuckets) else model, lr=lr, betas=betas,
                            weight_decay=weight_decay)
    if flat is None:
        flat = FLAT_ADAMW
    if flat and params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        from .optim import FlatAdamW
        return FlatAdamW(model, lr=lr, betas=betas, weight_decay=weight_decay, arena=arena)
    fused = params[0].is_cuda
    return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, betas=betas, fused=fused)


def training_step(model, batch, optimizer, arena=None, num_items=None):
    """One optimizer step on one micro-batch. Returns the (detached) loss tensor, no host sync."""
    if num_items is None:
        num_items = global_num_items(batch["labels"])
    out = model(**batch, num_items_in_batch=num_items)
    loss = out.loss
    loss.backward()
    if arena is not None:
        arena.finish()
    optimizer.step()
    if getattr(optimizer, "flat_p", None) is not None:
        optimizer.zero_grad()              # FlatAdamW: the step zeroed the gradient arena in its own pass
    elif arena is not None:
        arena.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=True)
    return loss.detach()


def unsloth_train(model, batches, optimizer=None, arena=None, max_steps=None, log_every=0):
    """trainer.py:49-57 counterpart for pre-tokenised batches (dicts with input_ids/labels[/position_ids/
    packed_seq_lengths] already on the model's device). Returns the list of per-step losses."""
    model.train()
    if arena is None and optimizer is not None and getattr(optimizer, "arena", None) is not None:
        arena = optimizer.arena                    # FlatAdamW built over (or with) an arena: that one exchanges the gradients
    base = model.get_base_model() if hasattr(model, "get_base_model") else model
    if arena is None and optimizer is None and getattr(base, "_unsloth_full_finetuning", False):
        optimizer = make_optimizer(model)          # ShardedAdamW owns its FullGradBuckets
        arena = optimizer.arena
    if arena is None and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        arena = LoRAGradArena(model)               # BEFORE the optimizer: FlatAdamW adopts it instead of creating a second one
    if optimizer is None:
        optimizer = make_optimizer(model, arena=arena)
    losses, t0 = [], time.time()
    for step, batch in enumerate(batches):
        if max_steps is not None and step >= max_steps:
            break
        losses.append(training_step(model, batch, optimizer, arena))
        if log_every and (step + 1) % log_every == 0:
            print(f"step {step + 1}: loss {float(losses[-1]):.4f}  ({time.time() - t0:.1f}s)", flush=True)
    return [float(l) for l in losses]


# trainer.py:988-1021 runs this at import time: scripts written for TRL < 0.13 (tokenizer=, config-only keyword arguments)
# work as soon as `unsloth_amd.trainer` is imported. No-op without TRL.
if HAS_TRL:
    _patch_trl_trainer()
