"""lora_tn (rank-16 weight gradients, csrc/lora_side.hip) on the step's problem sets; run under UNSLOTH_AMD_LIB=<build> for an
A/B of two builds (tools/lora_tn_ab_run.sh). One JSON line per problem set: us per launch pair (partials + reduce)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd.kernels import utils as U  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
DEV, bf = "cuda", torch.bfloat16


def timeit(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(ts)[2]


for T in (8192, 2048):
    H, I, Q, KV = 4096, 14336, 4096, 1024
    X = torch.randn(T, H, device=DEV, dtype=bf)
    Xi = torch.randn(T, I, device=DEV, dtype=bf)
    Xq = torch.randn(T, Q + 2 * KV, device=DEV, dtype=bf)
    P = torch.randn(T, 16, device=DEV)
    sets = {
        "MLP block (6 problems: 3 x 14336, 3 x 4096)": [(P, Xi, 16, False, 1.0), (P, X, 16, True, 1.0), (P, X, 16, False, 1.0),
                                                          (P, Xi, 16, True, 1.0), (P, X, 16, False, 1.0), (P, Xi, 16, True, 1.0)],
        "q|k|v (6 problems: 3 x 4096 in, 4096 / 1024 / 1024 out)": [(P, X, 16, False, 1.0), (P, Xq[:, :Q], 16, True, 1.0),
                                                                      (P, X, 16, False, 1.0), (P, Xq[:, Q:Q + KV], 16, True, 1.0),
                                                                      (P, X, 16, False, 1.0), (P, Xq[:, Q + KV:], 16, True, 1.0)],
        "o (2 problems: 4096, 4096)": [(P, X, 16, False, 1.0), (P, X, 16, True, 1.0)],
    }
    for name, probs in sets.items():
        outs = U.lora_tn(probs)
        us = timeit(lambda: U.lora_tn(probs))
        print(json.dumps(dict(lib=tag, T=T, problems=name, us=round(us, 2),
                              checksum=float(sum(o.float().abs().sum() for o in outs)))), flush=True)
