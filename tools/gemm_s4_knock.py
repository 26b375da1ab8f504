"""Variants and knock-outs of gemm_nt256s_kernel against the 8-wave kernel and torch.matmul, one process, interleaved.
Knock-outs need a library built with UAMD_EXTRA_CFLAGS=-DUAMD_G256S_KNOCKOUTS (their results are garbage: timing only).
    python tools/gemm_s4_knock.py [out.jsonl] [--knock]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import utils as U  # noqa: E402

DEV = "cuda"


def run(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = open(args[0], "w") if args else None
    knock = "--knock" in sys.argv
    bf = torch.bfloat16
    L = _lib.lib()
    shapes = [(8192, 4096, 4096, "o"), (8192, 4096, 14336, "down"), (8192, 28672, 4096, "gate+up"), (2048, 14336, 4096, "gate@2k")]
    if "--two" in sys.argv:
        shapes = shapes[:2]
    names = {0: "pp", 1: "s4"}
    if knock:
        names.update({3: "s4_no_dma", 4: "s4_no_reads", 5: "s4_mfma_sync", 6: "s4_mfma_only", 7: "s4_no_vmcnt", 8: "s4_no_barriers"})
    for M, N, K, tag in shapes:
        X = torch.randn(M, K, device=DEV, dtype=bf)
        W = (torch.randn(N, K, device=DEV) * 0.02).to(bf)
        ref = X @ W.t()
        cands = {"torch": lambda: X @ W.t()}

        def mk(v):
            def f():
                L.uamd_set_tuning(1, 8)
                L.uamd_set_tuning(11, v)
                U.GEMM256_MODE = "on"
                return U.lora_linear_forward(X, [(W, None, None, None, None)])[0]
            return f
        for v, n in names.items():
            cands[n] = mk(v)
        for name, f in cands.items():
            y = f()
            rel = float((y.float() - ref.float()).norm() / ref.float().norm())
            if name in ("pp", "s4") and not rel < 2e-2:
                print(json.dumps(dict(shape=tag, kernel=name, ERROR="mismatch", rel=rel)), flush=True)
        for f in cands.values():
            run(f, 3)
        best = {k: 1e9 for k in cands}
        for _ in range(5):
            for name, f in cands.items():
                best[name] = min(best[name], run(f, 10))
        fl = 2.0 * M * N * K
        rec = dict(shape=tag, M=M, N=N, K=K, tflops={k: round(fl / v / 1e12, 1) for k, v in best.items()},
                   us={k: round(v * 1e6, 1) for k, v in best.items()})
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
        L.uamd_set_tuning(11, 1)
        del X, W, ref


if __name__ == "__main__":
    main()
