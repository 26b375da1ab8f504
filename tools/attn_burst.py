"""Is the attention forward power / clock limited? Time single launches after an idle period against sustained launches."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd.kernels.attention import attn_forward
B, S, Hq, Hk, D = 4, 2048, 32, 8, 128
qkv = torch.randn(B, S, (Hq + 2 * Hk) * D, device="cuda", dtype=torch.bfloat16)
q = qkv[..., :Hq * D].view(B, S, Hq, D); k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, S, Hk, D); v = qkv[..., (Hq + Hk) * D:].view(B, S, Hk, D)
for _ in range(3): attn_forward(q, k, v)
torch.cuda.synchronize()
def one():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); attn_forward(q, k, v); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for idle in (0.0, 0.5, 2.0):
    res = []
    for rep in range(4):
        time.sleep(idle)
        res.append(round(one(), 1))
    print("idle %.1fs before each launch:" % idle, res)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): attn_forward(q, k, v)
e1.record(); torch.cuda.synchronize()
print("sustained 200 launches: %.1f us each" % (e0.elapsed_time(e1) * 1e3 / 200))
