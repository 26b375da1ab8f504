"""Shared by tests/test_hf_trainer_glue.py (CPU, compute replaced by a torch stand-in) and tests/test_gpu_hf_trainer.py (the
HIP path): drive a patched model through the STOCK `transformers.Trainer` -- the reference's real hot loop
(trainer.py:502-623 -> Trainer.training_step -> compute_loss -> PeftModel_fast_forward, models/_utils.py:3142-3313) --
and replay the very batches it consumed through a hand-written accumulation loop."""
import torch


class Docs(torch.utils.data.Dataset):
    """Pre-tokenised documents of mixed length."""

    def __init__(self, n, vocab, lo, hi, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.rows = [torch.randint(0, vocab, (int(torch.randint(lo, hi + 1, (1,), generator=g)),), generator=g).tolist()
                     for _ in range(n)]

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return {"input_ids": self.rows[i]}


class PaddingFreeCollator:
    """What TRL's padding-free collator hands the reference (trainer.py:903-912, utils/packing.py:241-284): the documents of a
    micro-batch in ONE row, position ids restarting, `packed_seq_lengths`. Remembers every batch it built, in order."""

    def __init__(self):
        self.seen = []

    def __call__(self, features):
        from unsloth_amd.utils.packing import enable_padding_free_metadata
        batch = enable_padding_free_metadata([f["input_ids"] for f in features])
        self.seen.append({k: v.clone() for k, v in batch.items()})
        return batch


class PaddedCollator:
    """Plain [B, T] rows (equal lengths), labels = ids: the case where a count over the unshifted labels is off by B."""

    def __init__(self, length):
        self.length = length
        self.seen = []

    def __call__(self, features):
        ids = torch.tensor([(f["input_ids"] * (1 + self.length // len(f["input_ids"])))[:self.length] for f in features])
        pos = torch.arange(self.length, dtype=torch.int32).unsqueeze(0).expand(len(features), -1).contiguous()
        batch = dict(input_ids=ids, labels=ids.clone(), position_ids=pos)
        self.seen.append({k: v.clone() for k, v in batch.items()})
        return batch


def shifted_targets(batch):
    return int((batch["labels"][..., 1:] != -100).sum())


def run_stock_trainer(model, dataset, collator, tmpdir, *, steps=3, accumulate=2, docs_per_micro_batch=3, bf16=False,
                      gradient_checkpointing=True, use_cpu=False, lr=2e-4):
    """Returns (logged losses per optimizer step, trainer)."""
    from transformers import Trainer, TrainingArguments
    args = TrainingArguments(
        output_dir=str(tmpdir), per_device_train_batch_size=docs_per_micro_batch, gradient_accumulation_steps=accumulate,
        max_steps=steps, learning_rate=lr, weight_decay=0.01, lr_scheduler_type="constant", warmup_steps=0, max_grad_norm=0.0,
        logging_steps=1, logging_strategy="steps", report_to=[], save_strategy="no", bf16=bf16,
        gradient_checkpointing=gradient_checkpointing, optim="adamw_torch", seed=11, remove_unused_columns=False,
        dataloader_num_workers=0, dataloader_pin_memory=False, use_cpu=use_cpu, disable_tqdm=True)
    trainer = Trainer(model=model, args=args, train_dataset=dataset, data_collator=collator)
    trainer.train()
    losses = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    return losses, trainer


def replay(model, batches, device, *, accumulate=2, optimizer=None, autocast=False):
    """The same optimizer steps by hand: sum over the window of (token-loss sum / num_items of the WHOLE window)."""
    out = []
    model.train()
    for s in range(0, len(batches) - accumulate + 1, accumulate):
        window = batches[s:s + accumulate]
        n = torch.tensor(sum(shifted_targets(b) for b in window), device=device)
        total = 0.0
        for b in window:
            b = {k: v.to(device) for k, v in b.items()}
            with torch.autocast(device_type=torch.device(device).type, dtype=torch.bfloat16, enabled=autocast):
                loss = model(**b, num_items_in_batch=n).loss
            loss.backward()
            total += float(loss.detach())
        optimizer.step()
        optimizer.zero_grad()
        out.append(total)
    return out
