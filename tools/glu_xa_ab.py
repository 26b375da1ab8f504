"""A/B of the fused gated-activation kernels (uamd_glu_fwd_xa_ws / uamd_glu_bwd_xa_ws) on one MI355X over UAMD_GLU_XA: 0 = 4 waves
per 16-row block (rounds 3-4), 1 = 8 waves, 2 = 8 waves + tiles requested two steps ahead, 3 = 2 + the columns of a row group split
in two over adjacent workgroups where the shipped shape rule says so, 8 = split always; interleaved rounds, min over
rounds, beside the plain activation kernels, cold and right after a burst of GEMMs. Llama-3-8B MLP widths, r = 16. TB/s =
ALGORITHMIC bytes (3 resp. 6 x [M, 14336] bf16) / time. JSON lines (appended to argv[1]); run it under UNSLOTH_AMD_LIB=<another
build> for library A/Bs (the label says which library)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import utils as U  # noqa: E402
U.GLU_FUSED = "all"
from unsloth_amd.kernels.swiglu import swiglu_DWf_DW_dfg_kernel, swiglu_fg_kernel  # noqa: E402

dev = torch.device("cuda", 0)
g_ = torch.Generator().manual_seed(0)
L = _lib.lib()
LIBTAG = os.path.basename(os.environ.get("UNSLOTH_AMD_LIB", "default"))


def run(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def heat(ms=60):
    """a burst of GEMMs: the activation kernels of a training step start at the clock a GEMM leaves behind"""
    a = torch.randn(8192, 4096, device=dev, dtype=torch.bfloat16)
    b = torch.randn(4096, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(int(ms / 0.25)):
        a @ b


out = open(sys.argv[1], "a") if len(sys.argv) > 1 else None
SHAPES = [tuple(int(x) for x in sh.split("x")) for sh in os.environ.get("GLU_AB_SHAPES", "8192x14336,4096x14336,2048x14336").split(",")]
for M, K in SHAPES:
    H, r = 4096, int(os.environ.get("GLU_AB_RANK", "16"))
    dt = torch.bfloat16
    e = torch.randn(M, K, generator=g_).to(dt).to(dev)
    g = torch.randn(M, K, generator=g_).to(dt).to(dev)
    DW = (torch.randn(M, K, generator=g_) * 0.1).to(dt).to(dev)
    mk = lambda o, i: ((torch.randn(o, i, generator=g_) * 0.02).to(dt).to(dev), None,
                       torch.nn.Parameter((torch.randn(r, i, generator=g_) * 0.02).to(dev)),
                       torch.nn.Parameter((torch.randn(o, r, generator=g_) * 0.02).to(dev)), 2.0)
    down, up, gate = mk(H, K), mk(K, H), mk(K, H)

    def knob(v, f):
        def go():
            L.uamd_set_tuning(10, v)
            return f()
        return go
    cands = {
        "fwd_xa0": knob(0, lambda: U.glu_fwd_xa("swiglu", e, g, down)),
        "fwd_xa1": knob(1, lambda: U.glu_fwd_xa("swiglu", e, g, down)),
        "fwd_xa2": knob(2, lambda: U.glu_fwd_xa("swiglu", e, g, down)),
        "fwd_xa3": knob(3, lambda: U.glu_fwd_xa("swiglu", e, g, down)),
        "fwd_xa8": knob(8, lambda: U.glu_fwd_xa("swiglu", e, g, down)),
        "fwd_plain": lambda: swiglu_fg_kernel(e, g),
        "bwd_xa0": knob(0, lambda: U.glu_bwd_terms("swiglu", DW, e, g, up, gate)),
        "bwd_xa1": knob(1, lambda: U.glu_bwd_terms("swiglu", DW, e, g, up, gate)),
        "bwd_xa2": knob(2, lambda: U.glu_bwd_terms("swiglu", DW, e, g, up, gate)),
        "bwd_xa3": knob(3, lambda: U.glu_bwd_terms("swiglu", DW, e, g, up, gate)),
        "bwd_xa8": knob(8, lambda: U.glu_bwd_terms("swiglu", DW, e, g, up, gate)),
        "bwd_plain": lambda: swiglu_DWf_DW_dfg_kernel(DW, e, g),
    }
    for f in cands.values():
        f()
    torch.cuda.synchronize()
    for hot in (False, True):
        best = {k: 1e9 for k in cands}
        for _ in range(5):
            for k, f in cands.items():
                if hot:
                    heat()
                best[k] = min(best[k], run(f, 6 if hot else 10))
        rec = dict(lib=LIBTAG, tokens=M, K=K, rank=r, after_gemm_burst=hot)
        for k, v in best.items():
            nbytes = (3 if k.startswith("fwd") else 6) * M * K * 2
            rec[k] = dict(us=round(v, 1), TBps=round(nbytes / v / 1e6, 2))
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
    L.uamd_set_tuning(10, 3)
