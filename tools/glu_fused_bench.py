"""Microbenchmark of the fused activation + LoRA skinny products against the separate launches they replace
(Llama-3-8B MLP widths, 8192 and 2048 tokens, r = 16). Prints JSON lines."""
import json
import sys
import os
import torch
sys.path.insert(0, ".")
from unsloth_amd.kernels import utils as U
U.GLU_FUSED = "all"      # the direct calls below must not be refused by the size / direction policy
from unsloth_amd.kernels.swiglu import swiglu_DWf_DW_dfg_kernel, swiglu_fg_kernel

dev = torch.device("cuda", 0)
g_ = torch.Generator().manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M in (8192, 2048):
    K, H, r = 14336, 4096, 16
    dt = torch.bfloat16
    e = torch.randn(M, K, generator=g_).to(dt).to(dev)
    g = torch.randn(M, K, generator=g_).to(dt).to(dev)
    DW = (torch.randn(M, K, generator=g_) * 0.1).to(dt).to(dev)
    mk = lambda o, i: ((torch.randn(o, i, generator=g_) * 0.02).to(dt).to(dev), None, torch.nn.Parameter((torch.randn(r, i, generator=g_) * 0.02).to(dev)),
                       torch.nn.Parameter((torch.randn(o, r, generator=g_) * 0.02).to(dev)), 2.0)
    down, up, gate = mk(H, K), mk(K, H), mk(K, H)
    U.glu_fwd_xa("swiglu", e, g, down)            # warm the factor casts
    f_fused = timeit(lambda: U.glu_fwd_xa("swiglu", e, g, down))

    def fwd_sep():
        h = swiglu_fg_kernel(e, g)
        U._xa_and_rank_block(h, [down[2]], True)
    f_sep = timeit(fwd_sep)
    b_fused = timeit(lambda: U.glu_bwd_terms("swiglu", DW, e, g, up, gate))

    def bwd_sep():
        swiglu_DWf_DW_dfg_kernel(DW, e, g)
        U.lora_dx_terms([e, g], [up, gate])
    b_sep = timeit(bwd_sep)
    gb_f, gb_b = 3 * M * K * 2 / 1e9, 6 * M * K * 2 / 1e9
    print(json.dumps(dict(tokens=M, fwd_fused_us=round(f_fused, 1), fwd_separate_us=round(f_sep, 1), bwd_fused_us=round(b_fused, 1),
                          bwd_separate_us=round(b_sep, 1), fwd_fused_TBps=round(gb_f / f_fused * 1e3, 2),
                          bwd_fused_TBps=round(gb_b / b_fused * 1e3, 2))), flush=True)
