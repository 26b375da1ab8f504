"""A/B of the 4-wave x 128x128 form of the 256-tile GEMM (csrc/gemm256_w4.inc, UAMD_TUNE_GEMM_W4) against the shipped
8-wave ping-pong kernel and hipBLASLt on the step's shapes (M = 8192 tokens), NT (forward) and NN (dX) forms, with the
LoRA rank block on the NT launches. First: results must be BIT-IDENTICAL to the 8-wave kernel."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import utils as U  # noqa: E402

DEV = "cuda"
KNOB_W4, KNOB_PERSIST = 8, 7


def run(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    bf = torch.bfloat16
    L = _lib.lib()
    U.GEMM256_MODE = "on"
    M = int(os.environ.get("W4_M", "8192"))
    shapes = [("o", [4096], 4096), ("qkv", [4096, 1024, 1024], 4096), ("gate+up", [14336, 14336], 4096), ("down", [4096], 14336)]
    for tag, Ns, K in shapes:
        X = torch.randn(M, K, device=DEV, dtype=bf)
        Ws = [(torch.randn(n, K, device=DEV) * 0.02).to(bf) for n in Ns]
        # rank block: XK [M, 64] (16 real columns per group), BK per group [N, 64]
        xk = torch.zeros(M, 64, device=DEV, dtype=bf)
        xk[:, :16 * len(Ns)] = torch.randn(M, 16 * len(Ns), device=DEV).to(bf)
        bks = []
        for gi, n in enumerate(Ns):
            b = torch.zeros(n, 64, device=DEV, dtype=bf)
            b[:, 16 * gi:16 * gi + 16] = (torch.randn(n, 16, device=DEV) * 0.02).to(bf)
            bks.append(b)
        outs = [torch.empty(M, n, device=DEV, dtype=bf) for n in Ns]

        def nt(w4, persist=1, rank=True):
            def f():
                L.uamd_set_tuning(KNOB_W4, w4)
                L.uamd_set_tuning(KNOB_PERSIST, persist)
                groups = [U._group(W, C, W.shape[0], W.stride(0), xk=xk if rank else None, bk=bk if rank else None)
                          for W, C, bk in zip(Ws, outs, bks)]
                U._launch_gemm(X, groups, nf4=False)
            return f
        # bit identity (rank block on and off)
        for rank in (True, False):
            nt(0, 0, rank)()
            ref = [o.clone() for o in outs]
            nt(1, 0, rank)()
            same = all(torch.equal(a, b) for a, b in zip(ref, outs))
            lib = torch.cat([X @ W.t() for W in Ws], 1)
            rel = float((torch.cat(ref, 1).float() - lib.float()).norm() / lib.float().norm()) if not rank else None
            print(json.dumps(dict(shape=tag, form="NT", rank_block=rank, bit_identical=same, rel_vs_hipblaslt=rel)), flush=True)
        cands = {"w8": nt(0), "w4": nt(1), "w8_norank": nt(0, 1, False), "w4_norank": nt(1, 1, False),
                 "hipblaslt": lambda: [X @ W.t() for W in Ws]}
        for f in cands.values():
            run(f, 3)
        best = {k: 1e9 for k in cands}
        for _ in range(5):
            for name, f in cands.items():
                best[name] = min(best[name], run(f, 10))
        fl = 2.0 * M * sum(Ns) * K
        print(json.dumps(dict(shape=tag, form="NT", M=M, N=sum(Ns), K=K, us={k: round(v * 1e6, 1) for k, v in best.items()},
                              tflops={k: round(fl / v / 1e12, 1) for k, v in best.items()})), flush=True)
        del X, Ws, outs
    # NN form: dX = dY [M, N] @ W [N, K]  (contract over the weight's rows)
    for tag, N, K in [("dX_down", 4096, 14336), ("dX_gate", 14336, 4096), ("dX_qkv", 6144, 4096), ("dX_o", 4096, 4096)]:
        dY = torch.randn(M, N, device=DEV, dtype=bf)
        W = (torch.randn(N, K, device=DEV) * 0.02).to(bf)
        out = torch.empty(M, K, device=DEV, dtype=bf)

        def nn(w4):
            def f():
                L.uamd_set_tuning(KNOB_W4, w4)
                U._launch_gemm(dY, [U._group(W, out, K, W.stride(0))], nf4=False, nn=True)
            return f
        nn(0)()
        ref = out.clone()
        nn(1)()
        print(json.dumps(dict(shape=tag, form="NN", bit_identical=bool(torch.equal(ref, out)))), flush=True)
        cands = {"w8": nn(0), "w4": nn(1), "hipblaslt": lambda: dY @ W}
        for f in cands.values():
            run(f, 3)
        best = {k: 1e9 for k in cands}
        for _ in range(5):
            for name, f in cands.items():
                best[name] = min(best[name], run(f, 10))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(shape=tag, form="NN", M=M, N=K, K=N, us={k: round(v * 1e6, 1) for k, v in best.items()},
                              tflops={k: round(fl / v / 1e12, 1) for k, v in best.items()})), flush=True)
        del dY, W, out
    L.uamd_set_tuning(KNOB_W4, 0)


if __name__ == "__main__":
    main()
