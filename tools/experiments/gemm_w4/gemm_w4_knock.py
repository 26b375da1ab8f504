"""Knock-out timing of the 4-wave GEMM on the long-K shape (down_proj): what does each ingredient of the K loop cost?"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib
from unsloth_amd.kernels import utils as U
from tools.gemm_w4_ab import run
L = _lib.lib(); U.GEMM256_MODE = "on"
M, N, K = 8192, 4096, 14336
X = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def mk(w4):
    def f():
        L.uamd_set_tuning(8, w4); L.uamd_set_tuning(7, 0)
        U._launch_gemm(X, [U._group(W, out, N, W.stride(0))], nf4=False)
    return f
names = {0: "3/3/2", 133: "3/3/2 (templ)", 142: "4/2/2", 151: "5/1/2", 160: "6/0/2", 150: "5/0/3", 141: "4/1/3", 152: "5/2/1", 161: "6/1/1", 140: "4/0/4", 143: "4/3/1", 132: "3/2/3", 123: "2/3/3"}
c = {v: mk(k) for k, v in names.items()}
for f in c.values(): run(f, 3)
best = {k: 1e9 for k in c}
for _ in range(4):
    for n, f in c.items(): best[n] = min(best[n], run(f, 10))
fl = 2.0 * M * N * K
print(json.dumps({k: [round(v * 1e6, 1), round(fl / v / 1e12, 1)] for k, v in best.items()}))
# correctness of the redistributed variants (their results must equal the shipped schedule's bit for bit)
mk(0)(); ref = out.clone()
for k in (151, 160, 123):
    mk(k)(); print(names[k], "bit-identical:", bool(torch.equal(ref, out)))
L.uamd_set_tuning(8, 0)
