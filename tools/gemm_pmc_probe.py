"""Launch each large-M GEMM kernel a few times on one shape so that `rocprofv3 --pmc ...` can attribute counters:
the 8-wave ping-pong kernel (knob 11 = 0), the one-wave-per-SIMD kernel (knob 11 = 1) and torch.matmul (hipBLASLt).
usage: rocprofv3 --kernel-trace --pmc <counters> -d out -- python tools/gemm_pmc_probe.py [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import utils as U  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (8192, 4096, 14336)
bf = torch.bfloat16
X = torch.randn(M, K, device="cuda", dtype=bf)
W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
L = _lib.lib()
U.GEMM256_MODE = "on"
for knob in (0, 1):
    L.uamd_set_tuning(11, knob)
    for _ in range(4):
        U.lora_linear_forward(X, [(W, None, None, None, None)])
    torch.cuda.synchronize()
L.uamd_set_tuning(11, 1)
for _ in range(4):
    X @ W.t()
torch.cuda.synchronize()
