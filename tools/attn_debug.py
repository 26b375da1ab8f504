import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_attention import ref_attention, g
from unsloth_amd.kernels.attention import attn_forward, attn_backward
B, T, Hq, Hk, D = [int(x) for x in sys.argv[1:5]] + [128]
dtype = torch.bfloat16
qkv = (torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(2)) * 1.0).to(dtype)
do = torch.randn(B, T, Hq, D, generator=g(3)).to(dtype)
scale = 1.0 / math.sqrt(D)
qr = qkv[..., :Hq * D].view(B, T, Hq, D).float().requires_grad_(True)
kr = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D).float().requires_grad_(True)
vr = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D).float().requires_grad_(True)
o_ref, _ = ref_attention(qr, kr, vr, scale)
o_ref.backward(do.float())
qd = qkv.cuda()
q = qd[..., :Hq * D].view(B, T, Hq, D); k = qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D); v = qd[..., (Hq + Hk) * D:].view(B, T, Hk, D)
o, lse = attn_forward(q, k, v, scale)
dq, dk, dv = attn_backward(do.cuda(), q, k, v, o, lse, scale)
for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
    e = (got.float().cpu() - want).abs()
    print(name, "max err", e.max().item(), "ref max", want.abs().max().item())
    per_t = e.amax(dim=(0, 2, 3))
    per_d = e.amax(dim=(0, 1, 2))
    print("  per key/pos (first 64):", [round(x, 2) for x in per_t[:64].tolist()])
    print("  per d:", [round(x, 2) for x in per_d.tolist()])
