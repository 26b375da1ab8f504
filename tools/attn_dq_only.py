"""dQ kernel alone at the bench shape (timing only; used for the knock-out builds of attn_bwd_dq4_kernel): the backward pair is
timed with the dK/dV kernel's time subtracted is NOT possible from Python, so this calls uamd_attn_bwd and reports the pair;
the knock-out deltas are deltas of the dQ kernel (the dK/dV kernel is the same in every build)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib
from unsloth_amd.kernels import attention as A
B, Hq, Hk, T, D = 4, 32, 8, 2048, 128
torch.manual_seed(0)
qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device="cuda", dtype=torch.bfloat16)
q = qkv[..., :Hq * D].view(B, T, Hq, D); k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D); v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
o, lse = A.attn_forward(q, k, v)
do = torch.randn_like(o)
L = _lib.lib()
def timed(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = {}
for knob in (0, 4):
    L.uamd_set_tuning(4, knob)
    for _ in range(3): A.attn_backward(do, q, k, v, o, lse)
    out[knob] = min(timed(lambda: A.attn_backward(do, q, k, v, o, lse), 10) for _ in range(5))
print(json.dumps(dict(tag=os.environ.get("TAG", ""), pair_ms_dq4=round(out[4], 4), pair_ms_dq_old=round(out[0], 4), dq4_minus_old_us=round((out[4] - out[0]) * 1e3, 1))))
