"""dump P / dS of (unit 0, key half 0, step 0) from a -DUAMD_ATTN_DEBUG build and compare with the fp32 reference"""
import math, sys, os, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_attention import g
from unsloth_amd import _lib
from unsloth_amd.kernels import attention as A
B, T, Hq, Hk, D = 1, 32, 1, 1, 128
dtype = torch.bfloat16
qkv = (torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(2)) * 1.0).to(dtype)
do = torch.randn(B, T, Hq, D, generator=g(3)).to(dtype)
scale = 1.0 / math.sqrt(D)
qd = qkv.cuda()
q = qd[..., :Hq * D].view(B, T, Hq, D); k = qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D); v = qd[..., (Hq + Hk) * D:].view(B, T, Hk, D)
o, lse = A.attn_forward(q, k, v, scale)
dod = do.cuda()
dq = torch.empty_like(q.contiguous()); dk = torch.empty(B, T, Hk, D, dtype=dtype, device="cuda"); dv = torch.empty_like(dk)
Tp = 32
delta = torch.zeros(B * Hq * Tp + 4096, dtype=torch.float32, device="cuda")
rc = _lib.lib().uamd_attn_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), _lib.ptr(dod), _lib.ptr(lse), _lib.ptr(dq), _lib.ptr(dk),
                              _lib.ptr(dv), _lib.ptr(delta), A._strides(q, k, v, o, dod, dq, dk, dv), B, T, Hq, Hk, D, Tp, float(scale), 1,
                              _lib.dtype_code(dtype), _lib.stream_of(q))
torch.cuda.synchronize()
dbg = delta[B * Hq * Tp:B * Hq * Tp + 2048].cpu().view(64, 2, 16)
# reference P[q][key], dS
qf, kf, vf, dof, of = (x.float().cpu() for x in (q[0, :, 0], k[0, :, 0], v[0, :, 0], dod[0, :, 0], o[0, :, 0]))
S = qf @ kf.t() * scale
mask = torch.ones(T, T, dtype=torch.bool).tril()
S = S.masked_fill(~mask, float("-inf"))
P = torch.softmax(S, -1)
dP = dof @ vf.t()
Dl = (dof * of).sum(-1, keepdim=True)
dS = P * (dP - Dl) * scale
Pk = torch.zeros(T, T); dSk = torch.zeros(T, T)
for lane in range(64):
    key, lh = lane & 31, lane >> 5
    for r in range(16):
        qi = (r & 3) + 8 * (r >> 2) + 4 * lh
        Pk[qi, key] = dbg[lane, 0, r]; dSk[qi, key] = dbg[lane, 1, r]
print("P err", (Pk - P).abs().max().item(), "dS err", (dSk - dS).abs().max().item(), "Pmax", P.max().item())
print("P row0..3 kernel:", Pk[:4, :6]); print("P row0..3 ref:", P[:4, :6])
print("delta kernel", delta[:8].cpu(), "ref", Dl[:8, 0]); print("lse", lse[0, 0, :4].cpu(), torch.logsumexp(S, -1)[:4])
