"""Time our flash attention forward (and backward when present) against torch SDPA at the bench shape."""
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd.kernels import attention as A  # noqa: E402

B, Hq, Hk, T, D = 4, 32, 8, 2048, 128
dev = "cuda"
bf = torch.bfloat16
qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device=dev, dtype=bf)
q = qkv[..., :Hq * D].view(B, T, Hq, D)
k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)


def run(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


fl = 4.0 * B * Hq * T * T * D / 2
t_ours = run(lambda: A.attn_forward(q, k, v))
qt, kt, vt = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
t_sdpa = run(lambda: F.scaled_dot_product_attention(qt, kt, vt, is_causal=True, enable_gqa=True))
o, _ = A.attn_forward(q, k, v)
o_ref = F.scaled_dot_product_attention(qt, kt, vt, is_causal=True, enable_gqa=True).transpose(1, 2)
print(json.dumps(dict(fwd_ms=round(t_ours, 3), fwd_TF=round(fl / t_ours / 1e9, 1), sdpa_fwd_ms=round(t_sdpa, 3),
                      sdpa_TF=round(fl / t_sdpa / 1e9, 1), max_diff_vs_sdpa=float((o.float() - o_ref.float()).abs().max()))))
if hasattr(A, "attn_backward"):
    o, lse = A.attn_forward(q, k, v)
    do = torch.randn_like(o)
    t_b = run(lambda: A.attn_backward(do, q, k, v, o, lse))
    qr, kr, vr = (x.detach().clone().requires_grad_(True) for x in (qt, kt, vt))

    def sdpa_fb():
        oo = F.scaled_dot_product_attention(qr, kr, vr, is_causal=True, enable_gqa=True)
        oo.backward(do.transpose(1, 2))
    t_fb = run(sdpa_fb)
    print(json.dumps(dict(bwd_ms=round(t_b, 3), bwd_TF=round(2.5 * fl / t_b / 1e9, 1), sdpa_fwd_bwd_ms=round(t_fb, 3))))
