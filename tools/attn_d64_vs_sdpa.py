"""head_dim 64 (TinyLlama / Llama-3.2-1B shapes): the head_dim-128 kernels on zero-padded heads against torch SDPA
(the library's flash kernel), forward + backward, plain causal. Decides the dispatch in models/llama.py:_attention."""
import json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd.kernels.attention import flash_attention
import torch.nn.functional as F

def run(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for (B, T, Hq, Hk, D) in [(4, 2048, 32, 4, 64), (4, 2048, 32, 8, 64), (1, 2048, 32, 4, 64), (4, 512, 32, 4, 64), (4, 2048, 28, 4, 128)]:
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, T, Hq, D, device="cuda", dtype=torch.bfloat16)
    def ours():
        q = qkv[..., :Hq * D].view(B, T, Hq, D); k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D); v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
        o = flash_attention(q, k, v)
        o.backward(do); qkv.grad = None
    def sdpa():
        q = qkv[..., :Hq * D].view(B, T, Hq, D).transpose(1, 2); k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D).transpose(1, 2)
        v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
        o.transpose(1, 2).backward(do); qkv.grad = None
    print(json.dumps(dict(B=B, T=T, Hq=Hq, Hk=Hk, D=D, ours_us=round(run(ours), 1), sdpa_us=round(run(sdpa), 1))), flush=True)
