"""Backward (dQ + dK/dV launches) timings on the bench's attention shapes, one JSON line; for library A/Bs through UNSLOTH_AMD_LIB.
    python tools/attn_bwd_time.py [tag]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd.kernels import attention as A  # noqa: E402


def timed(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def shape(B, T, Hq, Hk, docs=None, fwd=False):
    torch.manual_seed(0)
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * 128, device="cuda", dtype=torch.bfloat16)
    q = qkv[..., :Hq * 128].view(B, T, Hq, 128)
    k = qkv[..., Hq * 128:(Hq + Hk) * 128].view(B, T, Hk, 128)
    v = qkv[..., (Hq + Hk) * 128:].view(B, T, Hk, 128)
    band = A.attention_band(T, batch=B, seq_lengths=docs, device="cuda") if docs else None
    o, lse = A.attn_forward(q, k, v, None, band)
    do = torch.randn_like(o)
    f = (lambda: A.attn_forward(q, k, v, None, band)) if fwd else (lambda: A.attn_backward(do, q, k, v, o, lse, None, band))
    for _ in range(3):
        f()
    return round(sorted(timed(f) for _ in range(7))[3], 4)


g_ = torch.Generator().manual_seed(1)
lens, left = [], 8192
while left > 0:
    n = min(int(torch.randint(64, 2049, (1,), generator=g_)) // (1 if len(lens) % 3 == 0 else 4) or 64, left)
    n = max(n, min(64, left))
    lens.append(n)
    left -= n
fw = "fwd" in sys.argv[2:]
print(json.dumps({"lib": sys.argv[1] if len(sys.argv) > 1 else "", "what": "fwd" if fw else "bwd", "4x2048": shape(4, 2048, 32, 8, fwd=fw),
                  "1x2048": shape(1, 2048, 32, 8, fwd=fw), "2x4096": shape(2, 4096, 32, 8, fwd=fw),
                  "1x4096_32:4": shape(1, 4096, 32, 4, fwd=fw), "packed_1x8192": shape(1, 8192, 32, 8, lens, fwd=fw)}), flush=True)
