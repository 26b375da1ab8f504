"""20 backward launches at the primary shape (4 x 2048, 32:8 heads), for `rocprofv3 --kernel-trace --stats`: per-kernel time of
attn_bwd_dq_kernel / attn_bwd_dkdv4_kernel outside the step.    python tools/attn_bwd_prof.py [B T Hq Hk]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd.kernels import attention as A  # noqa: E402

B, T, Hq, Hk = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (4, 2048, 32, 8)
torch.manual_seed(0)
qkv = torch.randn(B, T, (Hq + 2 * Hk) * 128, device="cuda", dtype=torch.bfloat16)
q = qkv[..., :Hq * 128].view(B, T, Hq, 128)
k = qkv[..., Hq * 128:(Hq + Hk) * 128].view(B, T, Hk, 128)
v = qkv[..., (Hq + Hk) * 128:].view(B, T, Hk, 128)
o, lse = A.attn_forward(q, k, v)
do = torch.randn_like(o)
for _ in range(20):
    A.attn_forward(q, k, v)
    A.attn_backward(do, q, k, v, o, lse)
torch.cuda.synchronize()
