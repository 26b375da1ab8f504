"""Glue between a patched model and HuggingFace's stock `transformers.Trainer` -- the reference's hot loop is
`Trainer.train -> training_step -> compute_loss -> PeftModel_fast_forward` (trainer.py:502-623, models/_utils.py:3142-3313).

  mark_disable_data_parallel        models/_utils.py:244-249 (+ the `Trainer._wrap_model` patch, :187-241): one process
                                    that sees several GPUs must not have the LoRA model wrapped in torch.nn.DataParallel
                                    (the fused path keeps per-model state: rope tables, scratch, the gradient arena)
  patch_gradient_accumulation_fix   models/_utils.py:3200-3313: the loss is sum(token losses) / num_items_in_batch with
                                    num_items counted over the WHOLE accumulation window -- and counted over the shifted
                                    labels, which is what the fused linear-CE divides a window's sum by
  apply_accepts_loss_kwargs_fix     llama.py:2988: Trainer must hand `num_items_in_batch` to the model
  exclude_rope_inv_freq_from_ddp    models/loader_utils.py:849-865 (lives in unsloth_amd/dp.py; re-exported)

transformers 5.15 (installed here) already passes `num_items_in_batch` through `compute_loss` when the model's forward
takes **kwargs, so the reference's source-rewriting of `Trainer.training_step` (:3261-3313, written against 4.4x) has no
counterpart: what remains is the COUNT. `Trainer._get_num_items_in_batch` counts over `labels[..., 1:]` only when
`LOSS_MAPPING[model.loss_type] is ForCausalLMLoss` -- and `patch_loss_functions` (kernels/cross_entropy_loss.py) replaces
exactly that mapping entry, as the reference does (cross_entropy_loss.py:459-473): on a patched model the stock count
silently falls back to the unshifted labels, one extra target per row.
"""
import functools
import inspect

from ..dp import exclude_rope_inv_freq_from_ddp  # noqa: F401  (re-export: the reference imports it next to the others)


def iter_wrapped_models(model):
    """model, then everything it wraps through `.base_model` / `.model` / `.module`, each object once (a PEFT wrapper forwards
    unknown attributes to the wrapped model, so one chain alone can skip a level: all three are followed)."""
    seen, queue = set(), [model]
    while queue:
        m = queue.pop(0)
        if m is None or id(m) in seen or not hasattr(m, "__dict__"):
            continue
        seen.add(id(m))
        yield m
        for a in ("base_model", "model", "module"):
            try:
                queue.append(getattr(m, a, None))
            except Exception:
                pass


def is_unsloth_model(model):
    return any(getattr(m, "_unsloth_amd_fast", False) or getattr(m, "_unsloth_disable_data_parallel", False)
               for m in iter_wrapped_models(model))


def patch_trainer_data_parallel():
    """`Trainer._wrap_model` wraps in nn.DataParallel when `args.n_gpu > 1`; for a model carrying
    `_unsloth_disable_data_parallel` the call runs with `args._n_gpu = 1` (restored afterwards). Idempotent.
    Returns False when transformers has no Trainer / no `_wrap_model`."""
    try:
        from transformers.trainer import Trainer
    except Exception:
        return False
    wrap = getattr(Trainer, "_wrap_model", None)
    if wrap is None:
        return False
    if getattr(wrap, "_unsloth_data_parallel_patched", False):
        return True

    @functools.wraps(wrap)
    def wrap_model_one_device(self, model, *args, **kwargs):
        targs = getattr(self, "args", None)
        if (targs is None or not getattr(model, "_unsloth_disable_data_parallel", False)
                or getattr(model, "is_loaded_in_8bit", False) or getattr(targs, "n_gpu", 0) <= 1):
            return wrap(self, model, *args, **kwargs)
        missing = object()
        before = targs.__dict__.get("_n_gpu", missing)
        targs._n_gpu = 1
        try:
            return wrap(self, model, *args, **kwargs)
        finally:
            if before is missing:
                targs.__dict__.pop("_n_gpu", None)
            else:
                targs._n_gpu = before

    wrap_model_one_device._unsloth_data_parallel_patched = True
    wrap_model_one_device._unsloth_original_wrap_model = wrap
    Trainer._wrap_model = wrap_model_one_device
    return True


def mark_disable_data_parallel(model, disable=True):
    if disable:
        patch_trainer_data_parallel()
    for m in iter_wrapped_models(model):
        try:
            m._unsloth_disable_data_parallel = bool(disable)
        except Exception:
            pass
    return model


def apply_accepts_loss_kwargs_fix(model):
    """Trainer reads `model.accepts_loss_kwargs` before it inspects the forward's signature: say it outright."""
    for m in iter_wrapped_models(model):
        try:
            m.accepts_loss_kwargs = True
        except Exception:
            pass
    return model


def patch_gradient_accumulation_fix(Trainer=None):
    """Count `num_items_in_batch` over the SHIFTED labels for patched models (module docstring). Wraps
    `Trainer._get_num_items_in_batch` (transformers >= 4.50); on older layouts wraps `get_batch_samples` and recounts.
    Idempotent; returns the name of the method it wrapped, or None."""
    if Trainer is None:
        try:
            from transformers.trainer import Trainer
        except Exception:
            return None
    counter = getattr(Trainer, "_get_num_items_in_batch", None)
    if counter is not None:
        if getattr(counter, "_unsloth_shifted_count", False):
            return "_get_num_items_in_batch"

        @functools.wraps(counter)
        def count_shifted(self, batch_samples, device):
            if not is_unsloth_model(getattr(self, "model", None)) or getattr(self, "_loss_shifts_labels", False):
                return counter(self, batch_samples, device)
            had = "_loss_shifts_labels" in self.__dict__
            before = self.__dict__.get("_loss_shifts_labels")
            self._loss_shifts_labels = True
            try:
                return counter(self, batch_samples, device)
            finally:
                if had:
                    self._loss_shifts_labels = before
                else:
                    self.__dict__.pop("_loss_shifts_labels", None)

        count_shifted._unsloth_shifted_count = True
        Trainer._get_num_items_in_batch = count_shifted
        return "_get_num_items_in_batch"
    getter = getattr(Trainer, "get_batch_samples", None)
    if getter is None or getattr(getter, "_unsloth_shifted_count", False):
        return "get_batch_samples" if getter is not None else None
    n_params = len(inspect.signature(getter).parameters)

    @functools.wraps(getter)
    def _unsloth_get_batch_samples(self, epoch_iterator, num_batches, *rest):
        out = getter(self, epoch_iterator, num_batches, *rest[:max(0, n_params - 3)])
        batch_samples, n = out
        if n is not None and is_unsloth_model(getattr(self, "model", None)) and batch_samples \
                and "labels" in batch_samples[0]:
            n = sum((b["shift_labels"] if "shift_labels" in b else b["labels"][..., 1:]).ne(-100).sum() for b in batch_samples)
        return batch_samples, n

    _unsloth_get_batch_samples._unsloth_shifted_count = True
    Trainer.get_batch_samples = _unsloth_get_batch_samples
    return "get_batch_samples"


def prepare_for_trainer(model):
    """Everything above for one model; called from FastLlamaModel.post_load / patch_peft_model (llama.py:2988-3000, :3575-3595)."""
    apply_accepts_loss_kwargs_fix(model)
    patch_gradient_accumulation_fix()
    mark_disable_data_parallel(model)
    return exclude_rope_inv_freq_from_ddp(model)


# the reference's private names (models/_utils.py:97-98, loader_utils.py:849), for code that imports them
_mark_unsloth_disable_data_parallel = mark_disable_data_parallel
_patch_transformers_trainer_data_parallel = patch_trainer_data_parallel
_exclude_rope_inv_freq_from_ddp = exclude_rope_inv_freq_from_ddp
