"""A/B of the large-M GEMM kernels on one MI355X (one process, interleaved rounds, random N(0,1)*0.02-scale data):
torch.matmul (hipBLASLt), gemm256.hip's 8-wave kernels ("pp") and its one-wave-per-SIMD kernel ("s4"), contiguous and
row-padded operands. Checks each against torch first."""
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
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    outfile = out
    bf = torch.bfloat16
    L = _lib.lib()
    shapes = [(8192, 4096, 4096, "o"), (8192, 14336, 4096, "gate"), (8192, 4096, 14336, "down"),
              (8192, 28672, 4096, "gate+up"), (2048, 14336, 4096, "gate@2k"), (4096, 128256, 4096, "lm_head@4k")]
    for M, N, K, tag in shapes:
        X = torch.randn(M, K, device=DEV, dtype=bf)
        W = (torch.randn(N, K, device=DEV) * 0.02).to(bf)
        ref = X @ W.t()
        cands = {"torch": lambda: X @ W.t()}

        def mk(gm=8, s=0):
            def f():
                L.uamd_set_tuning(1, gm)
                L.uamd_set_tuning(11, s)
                U.GEMM256_MODE = "on"
                return U.lora_linear_forward(X, [(W, None, None, None, None)])[0]
            return f
        cands["pp"] = mk(8)            # the 8-wave ping-pong kernels (rounds 1-5)
        cands["s4"] = mk(8, 1)         # gemm_nt256s_kernel: one wave per SIMD, hand-ordered K loop (round 6)
        pad = int(os.environ.get("GEMM_AB_PAD", "64"))
        Xp = torch.empty(M, K + pad, device=DEV, dtype=bf)[:, :K]
        Xp.copy_(X)
        Wp = torch.empty(N, K + pad, device=DEV, dtype=bf)[:, :K]
        Wp.copy_(W)

        def pp_pad():
            L.uamd_set_tuning(1, 8)
            L.uamd_set_tuning(11, 1)
            U.GEMM256_MODE = "on"
            return U.lora_linear_forward(Xp, [(Wp, None, None, None, None)])[0]
        cands["s4_pad"] = pp_pad
        cands["torch_pad"] = lambda: Xp @ Wp.t()
        for name, f in cands.items():
            y = f()
            err = float((y.float() - ref.float()).abs().max())
            rel = float((y.float() - ref.float()).norm() / ref.float().norm())
            if rel > 2e-2 or err != err:
                print(json.dumps(dict(shape=tag, kernel=name, ERROR="mismatch", max_abs=err, rel=rel)), flush=True)
        for f in cands.values():
            run(f, 3)
        best = {k: 1e9 for k in cands}
        for _ in range(5):                      # interleaved rounds
            for name, f in cands.items():
                best[name] = min(best[name], run(f, 10))
        fl = 2.0 * M * N * K
        rec = dict(shape=tag, M=M, N=N, K=K, **{k: round(fl / v / 1e12, 1) for k, v in best.items()})
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
        del X, W, ref
    # the dX products (NN: B given as [K, N], the weight's own row-major layout): dY [M, out] @ W [out, in]
    from unsloth_amd.kernels.utils import _group, _launch_gemm
    for M, N, K, tag in [(8192, 14336, 4096, "down-dX"), (8192, 4096, 14336, "gate-dX"), (8192, 4096, 4096, "o-dX"),
                         (2048, 14336, 4096, "down-dX@2k")]:
        X = torch.randn(M, K, device=DEV, dtype=bf)
        W = (torch.randn(K, N, device=DEV) * 0.02).to(bf)
        ref = X @ W
        out = torch.empty(M, N, device=DEV, dtype=bf)

        def mkn(sv):
            def f():
                L.uamd_set_tuning(1, 8)
                L.uamd_set_tuning(11, sv)
                _launch_gemm(X, [_group(W, out, N, W.stride(0))], nf4=False, accumulate=False, nn=True)
                return out
            return f
        cands = {"torch": lambda: X @ W, "pp": mkn(0), "s4": mkn(1)}
        for name, f in cands.items():
            y = f()
            rel = float((y.float() - ref.float()).norm() / ref.float().norm())
            if rel > 2e-2 or rel != rel:
                print(json.dumps(dict(shape=tag, kernel=name, ERROR="mismatch", rel=rel)), flush=True)
        for f in cands.values():
            run(f, 3)
        best = {k: 1e9 for k in cands}
        for _ in range(5):
            for name, f in cands.items():
                best[name] = min(best[name], run(f, 10))
        fl = 2.0 * M * N * K
        rec = dict(shape=tag, form="NN", M=M, N=N, K=K, **{k: round(fl / v / 1e12, 1) for k, v in best.items()})
        print(json.dumps(rec), flush=True)
        if outfile:
            outfile.write(json.dumps(rec) + "\n")
            outfile.flush()
        L.uamd_set_tuning(11, 1)
        del X, W, ref


if __name__ == "__main__":
    main()
