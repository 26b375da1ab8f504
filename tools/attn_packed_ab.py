"""Forward attention on packed / windowed batches: one block per item (UAMD_TUNE_ATTN_VAR 1) vs the persistent kernel with the
static deal (10) vs the persistent kernel with claimed items (2), interleaved in one process.
    python tools/attn_packed_ab.py [arms]      default "1,10,2"
One JSON line per (shape, arm): ms (median of 7 rounds x 20 launches), algorithmic TFLOP/s, fraction of the 2.5 PFLOP/s peak."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import attention as A  # noqa: E402

ARMS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,10,2").split(",")]
dev, bf = "cuda", torch.bfloat16
L = _lib.lib()


def timed(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def chat_mix(total, seed):
    """bench.py's packed SFT batch: 64 ... 2048 tokens, two thirds of the documents a quarter as long."""
    g_ = torch.Generator().manual_seed(seed)
    lens, left = [], total
    while left > 0:
        n = int(torch.randint(64, 2049, (1,), generator=g_))
        n = min(n if int(torch.randint(0, 3, (1,), generator=g_)) == 0 else max(64, n // 4), left)
        lens.append(n)
        left -= n
    return lens


def shape(tag, B, Hq, Hk, T, docs=None, window=None):
    torch.manual_seed(0)
    D = 128
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device=dev, dtype=bf)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = A.attention_band(T, batch=B, seq_lengths=docs, sliding_window=window, device=dev)
    pairs = float((torch.arange(T, device=dev).unsqueeze(0) - band[0] + 1).sum())
    fl = 4.0 * D * Hq * pairs
    outs, res = {}, {a: [] for a in ARMS}
    for arm in ARMS:
        L.uamd_set_tuning(4, arm)
        for _ in range(3):
            outs[arm] = A.attn_forward(q, k, v, None, band)
    torch.cuda.synchronize()
    for _ in range(7):
        for arm in ARMS:
            L.uamd_set_tuning(4, arm)
            res[arm].append(timed(lambda: A.attn_forward(q, k, v, None, band)))
    L.uamd_set_tuning(4, 0)
    for arm in ARMS:
        t = sorted(res[arm])[len(res[arm]) // 2]
        rec = dict(shape=tag, arm=arm, fwd_ms=round(t, 4), fwd_TF=round(fl / t / 1e9, 1), fwd_frac=round(fl / t / 1e9 / 2500.0, 4),
                   max_abs_diff_vs_first_arm=float((outs[arm][0].float() - outs[ARMS[0]][0].float()).abs().max()))
        print(json.dumps(rec), flush=True)


for seed in (1, 2):
    lens = chat_mix(8192, seed)
    shape("1x8192 32:8 chat mix, %d documents" % len(lens), 1, 32, 8, 8192, docs=lens)
shape("1x8192 32:8 16 documents of 512", 1, 32, 8, 8192, docs=[512] * 16)
shape("1x8192 32:8 documents 4096 + 32 x 128", 1, 32, 8, 8192, docs=[4096] + [128] * 32)
shape("4x2048 32:8 chat mix (rows cut at 2048)", 4, 32, 8, 2048, docs=chat_mix(8192, 3))
shape("2x4096 32:8 sliding window 1024 (Mistral)", 2, 32, 8, 4096, window=1024)
shape("1x8192 28:4 chat mix (config 4 text tower)", 1, 28, 4, 8192, docs=chat_mix(8192, 4))
shape("1x16384 32:8 chat mix", 1, 32, 8, 16384, docs=chat_mix(16384, 5))
