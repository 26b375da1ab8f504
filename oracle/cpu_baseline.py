"""CPU baseline for bench.py -- TEST/MEASUREMENT INFRASTRUCTURE ONLY (never on the product path).

The reference has no CPU execution path for its kernels (tests/version_compat/test_trl_fake_train_cpu.py:
14-16); what it ships, and what BASELINE.md 2.1 names as the CPU baseline, is the torch-eager fp32
composition of the same math: HF-style RMSNorm, rotate-half RoPE (rope_embedding.py:402-432), torch SwiGLU
(swiglu.py:69-77), F.cross_entropy (llama.py:1545-1562), matmul_lora with W_quant=None (utils.py:1146-1170)
under torch autograd. This module times exactly that on the host cores for a BOUNDED sample: ONE decoder
layer (forward + backward, LoRA on all 7 projections) plus the lm_head + cross-entropy, at the benchmark
model's widths and a short token count, and extrapolates to tokens/s for the full depth.
"""
import os
import time

import torch
import torch.nn.functional as F


def usable_cores():
    """Threads this process may really use: affinity mask AND the cgroup CPU quota (os.cpu_count() reports the
    host's 256 hardware threads inside a quota-limited container, and oversubscribing them is 100x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def _rope(x, cos, sin):
    half = x.shape[-1] // 2
    rh = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return x * cos + rh * sin


def _lin(x, W, A, B, s):
    return x @ W.t() + s * ((x @ A.t()) @ B.t())


def time_layer(hidden=4096, inter=14336, n_heads=32, n_kv=8, head_dim=128, vocab=128256, r=16, tokens=256,
               n_layers=32, budget_s=25.0, seed=3407):
    torch.manual_seed(seed)
    threads = usable_cores()
    torch.set_num_threads(threads)
    f32 = torch.float32
    mk = lambda o, i: (torch.randn(o, i, dtype=f32) * 0.02, (torch.randn(r, i, dtype=f32) * 0.02).requires_grad_(True),
                       (torch.randn(o, r, dtype=f32) * 0.02).requires_grad_(True), 1.0)
    q, k, v, o = mk(n_heads * head_dim, hidden), mk(n_kv * head_dim, hidden), mk(n_kv * head_dim, hidden), mk(hidden, n_heads * head_dim)
    gate, up, down = mk(inter, hidden), mk(inter, hidden), mk(hidden, inter)
    w1, w2 = torch.ones(hidden), torch.ones(hidden)
    pos = torch.arange(tokens, dtype=f32)
    inv = 1.0 / (5e5 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    emb = torch.cat([torch.outer(pos, inv)] * 2, dim=-1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]
    lm_head = torch.randn(vocab, hidden, dtype=f32) * 0.02

    def norm(x, w):
        return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))

    def layer(h):
        x = norm(h, w1)
        Q = _lin(x, *q).view(1, tokens, n_heads, head_dim).transpose(1, 2)
        K = _lin(x, *k).view(1, tokens, n_kv, head_dim).transpose(1, 2)
        V = _lin(x, *v).view(1, tokens, n_kv, head_dim).transpose(1, 2)
        Q, K = _rope(Q, cos, sin), _rope(K, cos, sin)
        A = F.scaled_dot_product_attention(Q, K, V, is_causal=True, enable_gqa=True)
        h = h + _lin(A.transpose(1, 2).reshape(1, tokens, -1), *o)
        x = norm(h, w2)
        e, g = _lin(x, *gate), _lin(x, *up)
        return h + _lin(F.silu(e) * g, *down)

    h0 = torch.randn(1, tokens, hidden, dtype=f32).requires_grad_(True)
    labels = torch.randint(0, vocab, (tokens,))
    t_layer, t_head, reps = 0.0, 0.0, 0
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        out = layer(h0)
        out.sum().backward()
        t1 = time.perf_counter()
        hh = out.detach().requires_grad_(True)
        loss = F.cross_entropy((hh[0] @ lm_head.t())[:-1], labels[1:], reduction="sum") / (tokens - 1)
        loss.backward()
        t2 = time.perf_counter()
        t_layer += t1 - t0
        t_head += t2 - t1
        reps += 1
        if reps >= 1 and (time.perf_counter() - t_start) > budget_s * 0.6:
            break
    t_layer /= reps
    t_head /= reps
    step_s = t_layer * n_layers + t_head
    return dict(value=tokens / step_s, unit="tokens/s", cores=threads, kind="port",
                sample=(f"torch fp32 eager autograd on {threads} host threads: 1 decoder layer fwd+bwd "
                        f"({t_layer:.2f} s) + lm_head/CE ({t_head:.2f} s) at T={tokens} tokens, Llama-3-8B widths, "
                        f"LoRA r={r}; extrapolated x{n_layers} layers; {reps} repetition(s)"))


def time_config1(hidden=2048, inter=5632, n_layers=22, n_heads=32, n_kv=4, head_dim=64, vocab=32000, r=8, tokens=512,
                 budget_s=15.0, seed=3407):
    """BASELINE config 1, timed DIRECTLY: TinyLlama-1.1B widths, all 22 layers, LoRA r=8 on the 7
    projections, seq 512, batch 1, torch fp32 eager autograd (forward + backward, no optimizer: plumbing reference) on
    the host cores. Distinct weights per layer (4.4 GB of fp32), whole steps only -- nothing extrapolated."""
    torch.manual_seed(seed)
    threads = usable_cores()
    torch.set_num_threads(threads)
    f32 = torch.float32

    def mk(o, i):
        return (torch.empty(o, i, dtype=f32).normal_(0, 0.02), (torch.randn(r, i, dtype=f32) * 0.02).requires_grad_(True),
                (torch.randn(o, r, dtype=f32) * 0.02).requires_grad_(True), 1.0)
    layers = []
    for _ in range(n_layers):
        layers.append(dict(q=mk(n_heads * head_dim, hidden), k=mk(n_kv * head_dim, hidden), v=mk(n_kv * head_dim, hidden),
                           o=mk(hidden, n_heads * head_dim), gate=mk(inter, hidden), up=mk(inter, hidden),
                           down=mk(hidden, inter), w1=torch.ones(hidden), w2=torch.ones(hidden)))
    embed = torch.empty(vocab, hidden, dtype=f32).normal_(0, 0.02)
    lm_head = torch.empty(vocab, hidden, dtype=f32).normal_(0, 0.02)
    pos = torch.arange(tokens, dtype=f32)
    inv = 1.0 / (1e4 ** (torch.arange(0, head_dim, 2).float() / head_dim))
    emb = torch.cat([torch.outer(pos, inv)] * 2, dim=-1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]

    def norm(x, w):
        return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))

    def step(ids, depth=None):
        h = embed[ids][None]
        for L in layers[:depth]:
            x = norm(h, L["w1"])
            Q = _lin(x, *L["q"]).view(1, tokens, n_heads, head_dim).transpose(1, 2)
            K = _lin(x, *L["k"]).view(1, tokens, n_kv, head_dim).transpose(1, 2)
            V = _lin(x, *L["v"]).view(1, tokens, n_kv, head_dim).transpose(1, 2)
            Q, K = _rope(Q, cos, sin), _rope(K, cos, sin)
            A = F.scaled_dot_product_attention(Q, K, V, is_causal=True, enable_gqa=True)
            h = h + _lin(A.transpose(1, 2).reshape(1, tokens, -1), *L["o"])
            x = norm(h, L["w2"])
            h = h + _lin(F.silu(_lin(x, *L["gate"])) * _lin(x, *L["up"]), *L["down"])
        loss = F.cross_entropy((h[0] @ lm_head.t())[:-1], ids[1:], reduction="sum") / (tokens - 1)
        loss.backward()
        return float(loss.detach())

    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, vocab, (tokens,), generator=g)
    step(ids, 2)                                          # warm-up on two layers (allocator, thread pool)
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or (time.perf_counter() - t0) < budget_s * 0.5:
        step(ids)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return dict(value=round(tokens / dt, 2), unit="tokens/s", cores=threads, kind="port",
                sample=(f"BASELINE config 1: TinyLlama-1.1B widths, {n_layers} layers, LoRA r={r}, seq {tokens}, batch 1, torch "
                        f"fp32 eager autograd fwd+bwd on {threads} host threads, {reps} whole step(s) of {dt:.2f} s, nothing "
                        "extrapolated"))


if __name__ == "__main__":
    print(time_layer())
    print(time_config1())
