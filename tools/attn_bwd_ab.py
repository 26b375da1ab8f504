"""A/B of the attention backward at the bench shape (4 x 2048 tokens, 32:8 heads): dK/dV kernel of round 3 (4 waves x 64
keys, UAMD_TUNE_ATTN_VAR bit 1 clear) against round 1's (bit 1 set), interleaved in one process; error of each against
the fp64 gradients of a small slice. Prints one JSON line per arm + the difference between the arms."""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import attention as A  # noqa: E402

B, Hq, Hk, T, D = (int(x) for x in os.environ.get("ATTN_SHAPE", "4,32,8,2048,128").split(","))
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device=dev, dtype=bf)
q = qkv[..., :Hq * D].view(B, T, Hq, D)
k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
o, lse = A.attn_forward(q, k, v)
do = torch.randn_like(o)
L = _lib.lib()
fl = 4.0 * B * Hq * T * T * D / 2


def timed(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


ARMS = (0, 4, 2)
res = {k_: [] for k_ in ARMS}
outs = {}
for knob in ARMS:
    L.uamd_set_tuning(4, knob)
    outs[knob] = A.attn_backward(do, q, k, v, o, lse)
    for _ in range(3):
        A.attn_backward(do, q, k, v, o, lse)
torch.cuda.synchronize()
for rnd in range(6):
    for knob in ARMS:
        L.uamd_set_tuning(4, knob)
        res[knob].append(timed(lambda: A.attn_backward(do, q, k, v, o, lse), 10))
L.uamd_set_tuning(4, 0)
t_f = timed(lambda: A.attn_forward(q, k, v), 20)
print(json.dumps(dict(fwd_ms=round(t_f, 4), fwd_TF=round(fl / t_f / 1e9, 1))))
for knob, name in ((0, "dq (8 waves x 32 rows) + dkdv4"), (4, "dq4 + dkdv4 (one wave per SIMD)"), (2, "dq + dkdv (8 waves x 32 keys)")):
    ts = sorted(res[knob])
    print(json.dumps(dict(arm=name, bwd_ms_median=round(ts[len(ts) // 2], 4), bwd_ms_min=round(ts[0], 4),
                          bwd_TF_alg=round(2.5 * fl / ts[len(ts) // 2] / 1e9, 1))))
# agreement of the two arms + fp64 truth on one (batch, kv head) slice
d = {n: float((a.float() - b.float()).abs().max()) for n, a, b in zip(("dq", "dk", "dv"), outs[0], outs[4])}
print(json.dumps(dict(max_abs_diff_between_arms=d)))
Tn = min(T, 512)
qs, ks, vs = (x[:1, :Tn].double().detach().clone().requires_grad_(True) for x in (q[:, :, :Hq // Hk], k[:, :, :1], v[:, :, :1]))
s = torch.einsum("bthd,bshd->bhts", qs, ks.expand(-1, -1, Hq // Hk, -1)) / math.sqrt(D)
s = s.masked_fill(~torch.ones(Tn, Tn, dtype=torch.bool, device=dev).tril(), float("-inf"))
oo = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vs.expand(-1, -1, Hq // Hk, -1))
if Tn == T:
    oo.backward(do[:1, :Tn, :Hq // Hk].double())
    for knob in ARMS:
        dq_, dk_, dv_ = outs[knob]
        e = dict(dq=float((dq_[:1, :Tn, :Hq // Hk].double() - qs.grad).norm() / qs.grad.norm()),
                 dk=float((dk_[:1, :Tn, :1].double() - ks.grad).norm() / ks.grad.norm()),
                 dv=float((dv_[:1, :Tn, :1].double() - vs.grad).norm() / vs.grad.norm()))
        print(json.dumps(dict(knob=knob, rel_fro_vs_fp64=e)))
