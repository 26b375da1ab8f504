"""Measured error of the flash-attention kernels against the fp32 oracle on the parity-test cases, in the metrics the
tests bound: max |err|, relative Frobenius error, worst per-row relative error (row = one query's output / one token's
gradient). Output: one JSON line per (pass, dtype). tests/test_gpu_attention.py's tolerances are set from this."""
import json
import math
import sys

import torch

sys.path.insert(0, ".")
from tests.test_gpu_attention import g, ref_attention            # noqa: E402
from unsloth_amd.kernels.attention import attn_backward, attn_forward  # noqa: E402

CASES = [(1, 64, 4, 1), (2, 128, 8, 2), (1, 200, 4, 1), (1, 777, 8, 8), (2, 256, 8, 1), (1, 2048, 8, 2), (1, 31, 2, 1),
         (2, 1000, 4, 2), (1, 640, 4, 2)]


def metrics(got, want):
    d = got.float().cpu() - want
    row = d.flatten(0, -2).norm(dim=-1) / want.flatten(0, -2).norm(dim=-1).clamp_min(1e-20)
    return dict(max_abs=d.abs().max().item(), rel_fro=(d.norm() / want.norm()).item(), worst_row=row.max().item())


for dtype in (torch.bfloat16, torch.float16):
    worst = {}
    for (B, T, Hq, Hk) in CASES:
        D = 128
        qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(1)).to(dtype)
        do = torch.randn(B, T, Hq, D, generator=g(3)).to(dtype)
        scale = 1.0 / math.sqrt(D)
        qr = qkv[..., :Hq * D].view(B, T, Hq, D).float().requires_grad_(True)
        kr = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D).float().requires_grad_(True)
        vr = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D).float().requires_grad_(True)
        o_ref, lse_ref = ref_attention(qr, kr, vr, scale)
        o_ref.backward(do.float())
        qd = qkv.cuda()
        q = qd[..., :Hq * D].view(B, T, Hq, D)
        k = qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
        v = qd[..., (Hq + Hk) * D:].view(B, T, Hk, D)
        o, lse = attn_forward(q, k, v, scale)
        dq, dk, dv = attn_backward(do.cuda(), q, k, v, o, lse, scale)
        for name, got, want in (("o", o, o_ref.detach()), ("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
            m = metrics(got, want)
            w = worst.setdefault(name, dict(max_abs=0.0, rel_fro=0.0, worst_row=0.0))
            for kk in m:
                w[kk] = max(w[kk], m[kk])
        w = worst.setdefault("lse", dict(max_abs=0.0))
        w["max_abs"] = max(w["max_abs"], (lse.cpu() - lse_ref.detach()).abs().max().item())
    print(json.dumps(dict(dtype=str(dtype), worst=worst)))
