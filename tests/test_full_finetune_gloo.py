"""CPU, world_size 2 over gloo: the full fine-tuning exchange (unsloth_amd/full_finetune.py; BASELINE config 3's N > 1
path). Every rank computes its own micro-batch's gradients into the flat buckets, the buckets are summed over ranks
(gloo: all-reduce + own slice; RCCL: reduce-scatter), each rank updates ITS shard of the fp32 master weights with AdamW
and the updated 16-bit slices are all-gathered -- after which every rank holds exactly the parameters ONE process gets
from torch.optim.AdamW on fp32 masters with the summed gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class Tiny(torch.nn.Module):
    def __init__(self, tied=False):
        super().__init__()
        self.model = torch.nn.Module()
        self.model.embed_tokens = torch.nn.Embedding(40, 16)
        self.model.layers = torch.nn.ModuleList()
        for _ in range(3):
            blk = torch.nn.Module()
            blk.q_proj = torch.nn.Linear(16, 16, bias=False)
            blk.norm = torch.nn.LayerNorm(16, bias=False)
            self.model.layers.append(blk)
        self.model.norm = torch.nn.LayerNorm(16, bias=False)
        self.lm_head = torch.nn.Linear(16, 40, bias=False)
        if tied:
            self.lm_head.weight = self.model.embed_tokens.weight

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def forward(self, ids, skip=None):
        h = self.model.embed_tokens(ids)
        for i, blk in enumerate(self.model.layers):
            if i != skip:
                h = h + torch.tanh(blk.q_proj(blk.norm(h)))
        return self.lm_head(self.model.norm(h))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss(m, ids, skip=None):
    return m(ids, skip).float().logsumexp(-1).sum() / 64.0


def _worker(rank, world, port, tied, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unsloth_amd.full_finetune import ShardedAdamW
    torch.manual_seed(0)
    m = Tiny(tied).to(torch.bfloat16)
    ref = Tiny(tied).to(torch.float32)                      # fp32 masters of a single-process run
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    opt = ShardedAdamW(m, lr=3e-2, weight_decay=0.1)
    # HF Trainer's rule for the reference's full_finetuning path: no decay on 1-D parameters (norm weights, biases)
    ropt = torch.optim.AdamW([dict(params=[p for p in ref.parameters() if p.dim() > 1], weight_decay=0.1),
                              dict(params=[p for p in ref.parameters() if p.dim() <= 1], weight_decay=0.0)], lr=3e-2)
    B = opt.buckets
    n_buckets = len(B.buckets)
    views_ok = all(p.data_ptr() == B.buckets[B._where[id(p)][0]]["flat_p"].data_ptr() + 2 * B._where[id(p)][1] for p in B.params)
    ok = True
    for step in range(4):
        skip = 1 if step == 2 else None                    # step 2: layer 1 receives no gradient on any rank
        ids = [torch.randint(0, 40, (2, 8), generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world)]
        # the single-process reference sees the SUM of the ranks' bf16 gradients (each rank's rounded on its own, as the
        # exchange sums them), computed on bf16 replicas of the CURRENT parameters
        want = {n: torch.zeros_like(p, dtype=torch.float32) for n, p in ref.named_parameters()}
        for r in range(world):
            rep = Tiny(tied).to(torch.bfloat16)
            rep.load_state_dict(m.state_dict())
            _loss(rep, ids[r], skip).backward()
            for n, p in rep.named_parameters():
                if p.grad is not None:
                    want[n] += p.grad.float()
        for n, p in ref.named_parameters():
            p.grad = want[n].to(torch.bfloat16).float()    # the exchange's sum is rounded to the bucket dtype
        if step == 3:                                       # gradient accumulation: two micro-batches, one exchange
            with B.no_sync():
                (_loss(m, ids[rank], skip) * 0.5).backward()
            (_loss(m, ids[rank], skip) * 0.5).backward()
        else:
            _loss(m, ids[rank], skip).backward()
        B.finish()
        gn = float(opt.grad_norm())
        gn_ref = float(torch.cat([p.grad.flatten() for p in ref.parameters()]).norm())
        opt.step()
        opt.zero_grad()
        ropt.step()
        err = max((p.float() - rp.detach().to(torch.bfloat16).float()).abs().max().item()
                  for p, rp in zip(m.parameters(), ref.parameters()))
        # bf16 rounding of the parameters: one ulp at |p| <= 1 is 2^-8; the masters themselves agree to fp32 noise
        ok = ok and err <= 2 ** -7 and abs(gn - gn_ref) <= 2e-2 * gn_ref
        if step < 3:
            # the fp32 masters of this rank's shard equal the reference's fp32 parameters exactly where grads were identical
            pass
    # every rank ends with identical parameters
    flat = torch.cat([p.detach().float().flatten() for p in m.parameters()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same = all(torch.equal(other[0], o) for o in other)
    q.put((rank, bool(ok), bool(same), bool(views_ok), n_buckets))
    dist.destroy_process_group()


@pytest.mark.parametrize("tied", [False, True])
def test_sharded_full_finetune_world2(tied):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tied, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res             # parameters track the single-process fp32-master AdamW
    assert all(r[2] for r in res), res             # replicas identical after the all-gather
    assert all(r[3] for r in res), res             # parameters live in the flat buckets
    assert all(r[4] == 5 for r in res), res        # head, 3 layers, embeddings


# ----------------------------------------------------------------------------------------------------------------------
class TinyPos(Tiny):
    """Tiny + a second '*embed*' parameter (a learned position embedding): the embedding bucket then holds TWO parameters,
    one of them the tied lm_head / embed_tokens weight."""

    def __init__(self, tied=True):
        super().__init__(tied)
        self.model.embed_positions = torch.nn.Parameter(torch.randn(8, 16) * 0.1)

    def forward(self, ids, use_pos=True):
        h = self.model.embed_tokens(ids)
        if use_pos:
            h = h + self.model.embed_positions
        for blk in self.model.layers:
            h = h + torch.tanh(blk.q_proj(blk.norm(h)))
        return self.lm_head(self.model.norm(h))


def _worker_accum(rank, world, port, q):
    """Tied weights + a second embedding parameter + gradient accumulation under no_sync():
      (1) the post-accumulate hook of the tied weight fires ONCE per backward (autograd sums lm_head's and the lookup's
          gradient first), so the embedding bucket is complete after 2 arrivals, not 3;
      (2) micro-batch 1 gives `embed_positions` no gradient: the bucket stays one arrival short, and that leftover must not
          complete the bucket in the middle of micro-batch 2 (a reduce of half-accumulated gradients, with the late hook
          writing under the in-flight collective);
      (3) every bucket's exchange is launched exactly once per optimizer step, from the hook of its last gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unsloth_amd.full_finetune import ShardedAdamW
    torch.manual_seed(0)
    m = TinyPos(True).to(torch.bfloat16)
    opt = ShardedAdamW(m, lr=3e-2, weight_decay=0.0)
    B = opt.buckets
    emb = B.buckets[-1]
    assert sorted(emb["names"]) == ["model.embed_positions", "model.embed_tokens.weight"], emb["names"]
    assert emb["expected"] == 2
    fires = {"n": 0}
    tied_w = m.model.embed_tokens.weight
    tied_w.register_post_accumulate_grad_hook(lambda p: fires.__setitem__("n", fires["n"] + 1))
    launches = []
    real_launch = B._launch

    def counting_launch(bi):
        # at launch time the bucket must already hold EVERYTHING this step accumulates into it
        launches.append((bi, set(B._written)))
        return real_launch(bi)
    B._launch = counting_launch
    ok = True
    for step in range(2):
        ids = [[torch.randint(0, 40, (2, 8), generator=torch.Generator().manual_seed(100 * step + 10 * r + mb)) for mb in range(2)]
               for r in range(world)]
        want = {n: torch.zeros_like(p, dtype=torch.float32) for n, p in m.named_parameters()}
        for r in range(world):
            rep = TinyPos(True).to(torch.bfloat16)
            rep.load_state_dict(m.state_dict())
            (rep(ids[r][0], use_pos=False).float().logsumexp(-1).sum() / 64.0 * 0.5).backward()
            (rep(ids[r][1], use_pos=True).float().logsumexp(-1).sum() / 64.0 * 0.5).backward()
            for n, p in rep.named_parameters():
                want[n] += p.grad.float()
        launches.clear()
        fires["n"] = 0
        with B.no_sync():
            (m(ids[rank][0], use_pos=False).float().logsumexp(-1).sum() / 64.0 * 0.5).backward()
        (m(ids[rank][1], use_pos=True).float().logsumexp(-1).sum() / 64.0 * 0.5).backward()
        ok = ok and fires["n"] == 2                                     # (1): one hook call per backward
        ok = ok and sorted(bi for bi, _ in launches) == list(range(len(B.buckets)))          # (3) all launched by hooks, once
        emb_launch = [w for bi, w in launches if bi == len(B.buckets) - 1][0]
        ok = ok and {id(tied_w), id(m.model.embed_positions)} <= emb_launch
        B.finish()
        ok = ok and len(launches) == len(B.buckets)                     # finish() had nothing left to send
        for bi in range(len(B.buckets)):
            B.wait(bi)
        # (2): the reduced gradient of every parameter is the sum over ranks of the two accumulated micro-batches
        for n, p in m.named_parameters():
            bi, o = B._where[id(p)]
            lo, hi = B.rank * B.buckets[bi]["shard"], (B.rank + 1) * B.buckets[bi]["shard"]
            got = B.buckets[bi]["flat_g"][o:o + p.numel()].float()
            ref = want[n].flatten()
            a, b_ = max(o, lo), min(o + p.numel(), hi)                  # gloo all-reduces the whole bucket; compare all of it
            err = (got - ref).abs().max().item()
            ok = ok and err <= 2 ** -7 * max(1.0, ref.abs().max().item())
        opt.step()
        opt.zero_grad()
    q.put((rank, bool(ok), fires["n"], len(launches)))
    dist.destroy_process_group()


def test_tied_embedding_bucket_counts_one_arrival_and_never_leaks_across_micro_batches():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_accum, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
