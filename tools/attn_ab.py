"""Interleaved A/B of attention kernel variants (UAMD_TUNE_ATTN_VAR knob values) in ONE process, forward and backward, over the
shapes of the bench's operating points; error of every arm against an fp64 oracle on a slice.
    python tools/attn_ab.py [knob,knob,...] [fwd|bwd|fwd,bwd]      default "1,2": attn_fwd_kernel vs attn_fwd_ps_kernel
Prints one JSON line per (shape, arm): ms (median of 6 rounds x 10 launches), algorithmic TFLOP/s, fraction of the 2.5 PFLOP/s peak."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import attention as A  # noqa: E402

ARMS = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
WHAT = sys.argv[2] if len(sys.argv) > 2 else "fwd,bwd"
dev, bf = "cuda", torch.bfloat16
L = _lib.lib()


def timed(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def shape(tag, B, Hq, Hk, T, docs=None):
    torch.manual_seed(0)
    D = 128
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device=dev, dtype=bf)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = None
    pairs = B * T * (T + 1) / 2.0
    if docs is not None:
        band = A.attention_band(T, batch=B, seq_lengths=docs, device=dev)
        pairs = float((torch.arange(T, device=dev).unsqueeze(0) - band[0] + 1).sum())
    fl = 4.0 * D * Hq * pairs
    outs, res_f, res_b = {}, {a: [] for a in ARMS}, {a: [] for a in ARMS}
    for arm in ARMS:
        L.uamd_set_tuning(4, arm)
        o, lse = A.attn_forward(q, k, v, None, band)
        do = torch.randn_like(o) if arm == ARMS[0] else outs[ARMS[0]][2]
        g = A.attn_backward(do, q, k, v, o, lse, None, band) if "bwd" in WHAT else None
        outs[arm] = (o, lse, do, g)
        for _ in range(2):
            A.attn_forward(q, k, v, None, band)
    torch.cuda.synchronize()
    for _ in range(6):
        for arm in ARMS:
            L.uamd_set_tuning(4, arm)
            o, lse, do, _g = outs[arm]
            if "fwd" in WHAT:
                res_f[arm].append(timed(lambda: A.attn_forward(q, k, v, None, band)))
            if "bwd" in WHAT:
                res_b[arm].append(timed(lambda: A.attn_backward(do, q, k, v, o, lse, None, band)))
    L.uamd_set_tuning(4, 0)
    # fp64 oracle on (batch 0, kv head 0): all its query heads
    Gq = Hq // Hk
    Tn = min(T, 1024)
    qs = q[:1, :Tn, :Gq].double().detach().clone().requires_grad_(True)
    ks = k[:1, :Tn, :1].double().detach().clone().requires_grad_(True)
    vs = v[:1, :Tn, :1].double().detach().clone().requires_grad_(True)
    s = torch.einsum("bthd,bshd->bhts", qs, ks.expand(-1, -1, Gq, -1)) / math.sqrt(D)
    pos = torch.arange(Tn, device=dev)
    ok = pos[:, None] >= pos[None, :]
    if band is not None:
        ok = ok & (pos[None, :] >= band[0][0, :Tn, None])
    s = s.masked_fill(~ok, float("-inf"))
    oo = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vs.expand(-1, -1, Gq, -1))
    for arm in ARMS:
        o, lse, do, g = outs[arm]
        rec = dict(shape=tag, arm=arm)
        rec["o_rel_fro_vs_fp64"] = float((o[:1, :Tn, :Gq].double() - oo).norm() / oo.norm())
        if res_f[arm]:
            t = sorted(res_f[arm])[len(res_f[arm]) // 2]
            rec.update(fwd_ms=round(t, 4), fwd_TF=round(fl / t / 1e9, 1), fwd_frac=round(fl / t / 1e9 / 2500.0, 4))
        if res_b[arm]:
            t = sorted(res_b[arm])[len(res_b[arm]) // 2]
            rec.update(bwd_ms=round(t, 4), bwd_TF=round(2.5 * fl / t / 1e9, 1), bwd_frac=round(2.5 * fl / t / 1e9 / 2500.0, 4))
        if g is not None and Tn == T and band is None:
            if qs.grad is None:
                oo.backward(do[:1, :Tn, :Gq].double())
            rec["dq_dk_dv_rel_fro_vs_fp64"] = [float((a_[:1, :Tn, :n_].double() - r_.grad).norm() / r_.grad.norm())
                                               for a_, r_, n_ in ((g[0], qs, Gq), (g[1], ks, 1), (g[2], vs, 1))]
        if arm != ARMS[0]:
            rec["max_abs_diff_vs_first_arm"] = float((o.float() - outs[ARMS[0]][0].float()).abs().max())
        print(json.dumps(rec), flush=True)


shape("4x2048 32:8 (primary)", 4, 32, 8, 2048)
shape("1x2048 32:8 (batch 1)", 1, 32, 8, 2048)
shape("2x4096 32:8 (config 5)", 2, 32, 8, 4096)
shape("1x4096 32:4 (config 4: 28:4 padded to G=8)", 1, 32, 4, 4096)
g_ = torch.Generator().manual_seed(1)
lens, left = [], 8192
while left > 0:
    n = min(int(torch.randint(64, 2049, (1,), generator=g_)) // (1 if len(lens) % 3 == 0 else 4) or 64, left)
    n = max(n, min(64, left))
    lens.append(n)
    left -= n
shape("1x8192 32:8 packed, %d documents" % len(lens), 1, 32, 8, 8192, docs=lens)
shape("1x1024 32:8", 1, 32, 8, 1024)
