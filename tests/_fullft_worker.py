"""Worker of tests/test_gpu_dp_rccl.py::test_full_finetune_*: full fine-tuning steps on real RCCL (1 forced rank, or N
ranks with different batches): flat buckets, in-place reduce-scatter, sharded AdamW, in-place all-gather. Checks that
every rank ends with identical parameters, that they moved, and -- one rank -- that the exchange changed nothing
numerically against a run without any collective."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
    transport = "rccl"
    if os.environ.get("DP_TEST_ONE_DEVICE") == "1":            # two ranks on ONE GPU (see tests/_dp_worker.py)
        dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if os.environ.get("DP_TEST_BACKEND", "nccl") == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests._gloo_cuda_shim import install
        transport = "gloo " + install(dev)
    else:
        dist.init_process_group("nccl", device_id=dev)
    from transformers import LlamaConfig
    from unsloth_amd import FastLanguageModel
    from unsloth_amd.full_finetune import ShardedAdamW, full_finetune_step
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=128, vocab_size=2048, rms_norm_eps=1e-5, max_position_embeddings=1024,
                      rope_parameters={"rope_type": "default", "rope_theta": 5e5}, tie_word_embeddings=False)

    def build():
        m, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=256, full_finetuning=True, device=dev,
                                                 random_state=3407, use_gradient_checkpointing=False)
        return m
    def batch(r, step=0):
        gg = torch.Generator().manual_seed(500 + r + 10 * step)
        ids = torch.randint(0, 2048, (2, 256), generator=gg).to(dev)
        pos = torch.arange(256, dtype=torch.int32, device=dev).unsqueeze(0).expand(2, 256).contiguous()
        return dict(input_ids=ids, labels=ids.clone(), position_ids=pos)

    model = build()
    before = torch.cat([p.detach().float().flatten() for p in model.parameters()]).clone()
    opt = ShardedAdamW(model, lr=1e-3)
    assert opt.buckets._exchange, "the collectives must really be issued"
    if world > 1:
        # ---- the exchanged gradients against a single-process replay of EVERY rank's batch (plain autograd, no buckets, no
        #      collective). Both sides hold bf16 gradients: round(round(g_0) + round(g_1)) either way, up to the k order of
        #      the weight-gradient GEMMs -> compared at bf16 rounding noise (1e-2 of the norm), per parameter.
        from unsloth_amd.dp import global_num_items
        from unsloth_amd.kernels.utils import GRAD_SINKS
        FB0 = opt.buckets
        assert FB0._direct and all(GRAD_SINKS[i]() is FB0 for i in FB0._direct), "the GPU gradient sinks must be live"
        n_global = sum(int((batch(r)["labels"][:, 1:] != -100).sum()) for r in range(world))
        assert int(global_num_items(batch(rank)["labels"])) == n_global
        replay = build()
        for r in range(world):
            replay(**batch(r), num_items_in_batch=n_global).loss.backward()
        want = {n: p.grad.detach().float().clone() for n, p in replay.named_parameters()}
        del replay
        c0 = FB0.collectives
        model(**batch(rank), num_items_in_batch=n_global).loss.backward()
        FB0.finish()
        assert FB0.collectives - c0 == len(FB0.buckets)
        worst = 0.0
        for bi, b in enumerate(FB0.buckets):
            FB0.wait(bi)
            lo, hi = (0, b["numel"]) if FB0._gloo else (FB0.rank * b["shard"], (FB0.rank + 1) * b["shard"])
            for n, p, o in zip(b["names"], b["params"], b["offsets"]):
                a, e = max(o, lo), min(o + p.numel(), hi)          # the part of this parameter whose SUM this rank holds
                if a >= e:
                    continue
                got = b["flat_g"][a:e].float()
                ref = want[n].reshape(-1)[a - o:e - o]
                err = float((got - ref).norm() / (ref.norm() + 1e-30))
                worst = max(worst, err)
                assert err < 1e-2, f"rank {rank}: exchanged gradient of {n} differs from the replay of both batches: {err}"
        opt.zero_grad()
        print(f"rank {rank}: exchanged gradients == replay of {world} batches, worst rel err {worst:.2e}", flush=True)
    g = torch.Generator().manual_seed(100 + rank)
    losses = []
    # every bucket is exchanged exactly ONCE per step, and only when every gradient that belongs to it has been produced
    # (round 4: counted twice -- sink report + torch's hook -- a bucket was reduce-scattered when half complete, then again)
    FB = opt.buckets
    real_launch = FB._launch

    def checked_launch(bi):
        missing = [n for n, p in zip(FB.buckets[bi]["names"], FB.buckets[bi]["params"]) if id(p) not in FB._written]
        assert not missing, f"bucket {bi} exchanged before the gradients of {missing[:3]} were produced"
        return real_launch(bi)
    FB._launch = checked_launch
    for step in range(3):
        ids = torch.randint(0, 2048, (2, 256), generator=g).to(dev)
        pos = torch.arange(256, dtype=torch.int32, device=dev).unsqueeze(0).expand(2, 256).contiguous()
        c0 = FB.collectives
        losses.append(float(full_finetune_step(model, dict(input_ids=ids, labels=ids.clone(), position_ids=pos), opt)))
        assert FB.collectives - c0 == len(FB.buckets), (FB.collectives - c0, len(FB.buckets))
    after = torch.cat([p.detach().float().flatten() for p in model.parameters()])
    assert torch.isfinite(after).all() and float((after - before).abs().max()) > 0
    # replicas identical
    mine = after.double().sum().reshape(1)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    assert all(torch.equal(allv[0], v) for v in allv), allv
    if world == 1:
        # the same three steps without any collective (the environment switch off): bit-identical parameters
        opt.buckets.close()
        os.environ["UNSLOTH_AMD_DP_FORCE"] = "0"
        model2 = build()
        opt2 = ShardedAdamW(model2, lr=1e-3)
        assert not opt2.buckets._exchange
        g2 = torch.Generator().manual_seed(100 + rank)
        for step in range(3):
            ids = torch.randint(0, 2048, (2, 256), generator=g2).to(dev)
            pos = torch.arange(256, dtype=torch.int32, device=dev).unsqueeze(0).expand(2, 256).contiguous()
            full_finetune_step(model2, dict(input_ids=ids, labels=ids.clone(), position_ids=pos), opt2)
        after2 = torch.cat([p.detach().float().flatten() for p in model2.parameters()])
        assert torch.equal(after, after2), float((after - after2).abs().max())
    print(f"rank {rank}/{world} ok [{transport}] losses {losses}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
