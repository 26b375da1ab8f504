"""attn_bwd_dkdv4_kernel: the generated step loop (csrc/attn_kd4_loop.inc) against the C++ step body of the same kernel
(UAMD_TUNE_ATTN_VAR bit 2: every step through the C++ body). Same arithmetic, same order -> dK / dV must be BIT-IDENTICAL;
then both against an fp64 oracle on a small shape, and timings.
    python tools/attn_kd4_check.py [time]"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import attention as A  # noqa: E402

dev = "cuda"
L = _lib.lib()


def timed(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def case(tag, B, T, Hq, Hk, dtype=torch.bfloat16, docs=None, causal=True, time_it=False):
    torch.manual_seed(1)
    D = 128
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device=dev, dtype=dtype)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = None
    if docs is not None:
        band = (A.attention_band if causal else A.document_band)(T, batch=B, seq_lengths=docs, device=dev)
    o, lse = A.attn_forward(q, k, v, None, band, causal)
    do = torch.randn_like(o)
    res = {}
    for arm in (0, 4):
        L.uamd_set_tuning(4, arm)
        res[arm] = [t.clone() for t in A.attn_backward(do, q, k, v, o, lse, None, band, causal)]
    L.uamd_set_tuning(4, 0)
    torch.cuda.synchronize()
    rec = dict(case=tag, dtype=str(dtype).split(".")[-1])
    for name, a, b in zip(("dq", "dk", "dv"), res[0], res[4]):
        rec[name + "_bitwise"] = bool(torch.equal(a, b))
        if not rec[name + "_bitwise"]:
            rec[name + "_maxdiff"] = float((a.float() - b.float()).abs().max())
            rec[name + "_nan"] = bool(torch.isnan(a.float()).any())
    if T <= 1024 and docs is None and causal:
        G = Hq // Hk
        qs = q[:1, :, :G].double().detach().clone().requires_grad_(True)
        ks = k[:1, :, :1].double().detach().clone().requires_grad_(True)
        vs = v[:1, :, :1].double().detach().clone().requires_grad_(True)
        s = torch.einsum("bthd,bshd->bhts", qs, ks.expand(-1, -1, G, -1)) / math.sqrt(D)
        pos = torch.arange(T, device=dev)
        s = s.masked_fill(~(pos[:, None] >= pos[None, :]), float("-inf"))
        oo = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vs.expand(-1, -1, G, -1))
        oo.backward(do[:1, :, :G].double())
        rec["rel_fro_vs_fp64(dq,dk,dv)"] = [round(float((a_[:1, :, :n_].double() - r_.grad).norm() / r_.grad.norm()), 6)
                                            for a_, r_, n_ in ((res[0][0], qs, G), (res[0][1], ks, 1), (res[0][2], vs, 1))]
    if time_it:
        for arm in (0, 4):
            L.uamd_set_tuning(4, arm)
            for _ in range(3):
                A.attn_backward(do, q, k, v, o, lse, None, band, causal)
            ts = sorted(timed(lambda: A.attn_backward(do, q, k, v, o, lse, None, band, causal)) for _ in range(5))
            rec[f"bwd_ms_arm{arm}"] = round(ts[2], 4)
        L.uamd_set_tuning(4, 0)
    print(json.dumps(rec), flush=True)


tm = len(sys.argv) > 1 and sys.argv[1] == "time"
case("1x1024 32:8", 1, 1024, 32, 8, time_it=tm)
case("1x1024 32:8 f16", 1, 1024, 32, 8, dtype=torch.float16)
case("4x2048 32:8 (primary)", 4, 2048, 32, 8, time_it=tm)
case("1x2048 32:8", 1, 2048, 32, 8, time_it=tm)
case("2x4096 32:8", 2, 4096, 32, 8, time_it=tm)
case("1x1000 32:8 ragged", 1, 1000, 32, 8)
case("2x777 8:8 G=1", 2, 777, 8, 8)
case("1x2048 16:8 G=2", 1, 2048, 16, 8)
case("1x1024 64:8 G=8", 1, 1024, 64, 8)
case("1x4096 32:4 G=8", 1, 4096, 32, 4, time_it=tm)
g_ = torch.Generator().manual_seed(1)
lens, left = [], 8192
while left > 0:
    n = min(int(torch.randint(64, 2049, (1,), generator=g_)) // (1 if len(lens) % 3 == 0 else 4) or 64, left)
    n = max(n, min(64, left))
    lens.append(n)
    left -= n
case("1x8192 32:8 packed %d docs" % len(lens), 1, 8192, 32, 8, docs=lens, time_it=tm)
case("1x4096 16:16 non-causal windows", 1, 4096, 16, 16, docs=[1024, 1024, 1000, 1048], causal=False)
