"""Worker of tests/test_gpu_dp_rccl.py: one process per rank, RCCL ("nccl" backend) over the visible GPUs -- or, with
DP_TEST_BACKEND=gloo and DP_TEST_ONE_DEVICE=1, TWO ranks sharing device 0 (RCCL refuses two ranks on one device; gloo carries
the collectives, natively or host-staged by tests/_gloo_cuda_shim.py): the one configuration that ever shipped wrong
numbers here -- REAL GPU gradient sinks (kernels/utils.GRAD_SINKS: the fused LoRA-gradient kernel adds straight into the
arena) x world_size > 1 -- on the single GPU a test box has.
Every rank builds the same tiny QLoRA model, trains one step on ITS batch through dp.LoRAGradArena (bucketed async
all-reduce launched from the gradient hooks, overlapped with the backward), and checks the reduced gradients
against a local replay of every rank's batch without any collective; then gradient accumulation under no_sync(), a step
after zero_grad(set_to_none=True), and the optimizer step (replicas must end identical). Exit code 0 = pass."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build(dev, gc):
    from transformers import LlamaConfig
    from unsloth_amd import FastLanguageModel
    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=128, vocab_size=1000, rms_norm_eps=1e-5,
                      max_position_embeddings=256, rope_parameters={"rope_type": "default", "rope_theta": 5e5},
                      tie_word_embeddings=False)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=128, load_in_4bit=True, device=dev,
                                                 random_state=3407, use_gradient_checkpointing=gc)
    model = FastLanguageModel.get_peft_model(model, r=8, lora_alpha=16, use_gradient_checkpointing=gc, random_state=3407)
    g = torch.Generator().manual_seed(1)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=g) * 0.05).to(dev))
    return model


def batch_of(rank, dev, micro=0):
    g = torch.Generator().manual_seed(100 + rank + 1000 * micro)
    ids = torch.randint(0, 1000, (2, 64), generator=g)
    labels = ids.clone()
    labels[0, : 3 + rank] = -100
    pos = torch.arange(64, dtype=torch.int32).unsqueeze(0).expand(2, 64).contiguous()
    return dict(input_ids=ids.to(dev), labels=labels.to(dev), position_ids=pos.to(dev))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("DP_TEST_BACKEND", "nccl")
    dev = torch.device("cuda", 0 if os.environ.get("DP_TEST_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", rank)))
    torch.cuda.set_device(dev)
    transport = "rccl"
    if backend == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests._gloo_cuda_shim import install
        transport = "gloo " + install(dev)
    else:
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    from unsloth_amd.dp import LoRAGradArena, global_num_items
    from unsloth_amd.kernels.utils import GRAD_SINKS
    gc = os.environ.get("DP_TEST_GC", "off")
    gc = {"off": False, "on": True}.get(gc, gc)
    # ---- reference: every rank's batch replayed locally, loss normalised by the GLOBAL token count, grads summed
    ref = build(dev, gc)
    n_global = sum(int((batch_of(r, dev)["labels"][:, 1:] != -100).sum()) for r in range(world))
    for r in range(world):
        out = ref(**batch_of(r, dev), num_items_in_batch=n_global)
        out.loss.backward()
    want = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.requires_grad}
    del ref
    # ---- data parallel: this rank's batch only, exchange through the arena
    model = build(dev, gc)
    arena = LoRAGradArena(model, bucket_bytes=int(os.environ.get("DP_TEST_BUCKET", 1 << 16)))
    # REAL GPU sinks: every trainable factor's gradient is added into the arena by the fused kernel, not by AccumulateGrad
    assert arena.arena.is_cuda and all(GRAD_SINKS.get(id(p)) is not None and GRAD_SINKS[id(p)]() is arena for p in arena.params)
    mine = batch_of(rank, dev)
    n = global_num_items(mine["labels"])
    assert int(n) == n_global, (int(n), n_global)
    # every bucket is all-reduced exactly ONCE per step and only when all its gradients are in (round 4: the sink report plus
    # torch's post-accumulate hook counted every parameter twice -- two exchanges per bucket, the first one half complete)
    real_launch = arena._launch

    def checked_launch(b):
        members = [p for p in arena.params if arena._bucket_of[id(p)] == b]
        assert all(id(p) in arena._arrived for p in members) or not arena.overlap, f"bucket {b} exchanged while incomplete"
        return real_launch(b)
    arena._launch = checked_launch
    for step in range(2):                     # second step: the arena was zeroed, views kept
        c0 = arena.collectives
        out = model(**mine, num_items_in_batch=n)
        out.loss.backward()
        hooked = arena.collectives - c0       # launched from the gradient hooks, inside the backward
        arena.finish()
        assert arena.collectives - c0 == len(arena.buckets), (arena.collectives - c0, len(arena.buckets))
        assert hooked == len(arena.buckets), "every bucket's exchange starts inside the backward, from its last gradient"
        worst = 0.0
        for name, p in model.named_parameters():
            if p.requires_grad:
                err = float((p.grad - want[name]).norm() / (want[name].norm() + 1e-30))
                worst = max(worst, err)
        assert worst < 1e-5, f"rank {rank} step {step}: reduced gradient differs from the local replay: {worst}"
        arena.zero_grad()

    def check(tag, want_):
        w = max(float((p.grad - want_[name]).norm() / (want_[name].norm() + 1e-30))
                for name, p in model.named_parameters() if p.requires_grad)
        assert w < 1e-5, f"rank {rank} {tag}: reduced gradient differs from the local replay: {w}"
        return w
    # ---- after zero_grad(set_to_none=True) (PyTorch's default): the sinks re-attach the arena views and start from zero,
    #      although the slices still hold the previous step's REDUCED sums
    out = model(**mine, num_items_in_batch=n)
    out.loss.backward()
    arena.finish()
    for p in arena.params:
        p.grad = None
    arena.reset_arrivals()
    c0 = arena.collectives
    out = model(**mine, num_items_in_batch=n)
    out.loss.backward()
    arena.finish()
    assert arena.collectives - c0 == len(arena.buckets)
    worst = max(worst, check("after set_to_none", want))
    arena.zero_grad()
    # ---- gradient accumulation: micro-batch 0 under no_sync() (no exchange), micro-batch 1 exchanges the SUM of both
    ref = build(dev, gc)
    n_acc = sum(int((batch_of(r, dev, m)["labels"][:, 1:] != -100).sum()) for r in range(world) for m in range(2))
    for r in range(world):
        for m in range(2):
            ref(**batch_of(r, dev, m), num_items_in_batch=n_acc).loss.backward()
    want_acc = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.requires_grad}
    del ref
    c0 = arena.collectives
    with arena.no_sync():
        model(**batch_of(rank, dev, 0), num_items_in_batch=n_acc).loss.backward()
    assert arena.collectives == c0, "no exchange inside no_sync()"
    model(**batch_of(rank, dev, 1), num_items_in_batch=n_acc).loss.backward()
    arena.finish()
    assert arena.collectives - c0 == len(arena.buckets), "one exchange per bucket for the whole accumulation window"
    worst = max(worst, check("accumulation under no_sync", want_acc))
    arena.zero_grad()
    # ---- the whole training step on top of the same arena: optim.FlatAdamW steps on the REDUCED gradients in one
    #      launch and zeroes the arena in the same pass; every rank must end with identical parameters
    from unsloth_amd.optim import FlatAdamW
    from unsloth_amd.trainer import make_optimizer, training_step
    opt = make_optimizer(model, lr=1e-3, arena=arena)
    assert isinstance(opt, FlatAdamW) and opt.arena is arena
    before = opt.flat_p.clone()
    for step in range(2):
        training_step(model, mine, opt, arena, n)
        assert float(arena.arena.abs().max()) == 0.0, "the optimizer step leaves the gradient arena zeroed"
    assert float((opt.flat_p - before).abs().max()) > 0.0
    mx, mn = opt.flat_p.clone(), opt.flat_p.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    assert torch.equal(mx, mn), "replicas diverged after the optimizer step"
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    assert int(one.item()) == world
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok [{transport}]: buckets {arena.describe()['buckets']}, worst rel err {worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
