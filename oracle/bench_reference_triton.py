"""TEST INFRASTRUCTURE (oracle/): times the REFERENCE's own Triton kernels natively on the MI355X.

Same shapes and byte counts as tools/microbench.py (Llama-3-8B widths, T tokens), so the two JSONL files line up:
the reference's Triton RMSNorm / RoPE / SwiGLU / CE on triton-rocm next to our HIP kernels, same box.
Runs on the GPU box from the modules staged by oracle/stage_reference.py. Never imported by the product.

    gpurun -- 'python oracle/bench_reference_triton.py --out gpurun_out/ref_triton_bench.jsonl'
"""
import argparse
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_bf16_gpu import _find_reference, load_reference  # noqa: E402

HBM_PEAK = 8000.0


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--tokens", type=int, nargs="+", default=[2048, 8192])
    a = ap.parse_args()
    R = load_reference(_find_reference())
    rms, rope, ce, sw = R["rms_layernorm"], R["rope_embedding"], R["cross_entropy_loss"], R["swiglu"]
    out = open(a.out, "w") if a.out else None
    dev, bf = "cuda", torch.bfloat16
    H, I, V, Hq, Hk, D = 4096, 14336, 128256, 32, 8, 128

    def emit(name, secs, nbytes, **kw):
        rec = dict(kernel="ref_triton:" + name, us=round(secs * 1e6, 2), GBps=round(nbytes / secs / 1e9, 1),
                   frac_hbm=round(nbytes / secs / 1e9 / HBM_PEAK, 3), **kw)
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()

    for T in a.tokens:
        try:
            X = torch.randn(T, H, device=dev, dtype=bf)
            W = torch.rand(H, device=dev, dtype=bf)
            emit("rms_fwd", timeit(lambda: rms.Fast_RMS_Layernorm.apply(X, W, 1e-5, False)), 2 * T * H * 2 + H * 2 + T * 4, T=T)
            Xg = X.clone().requires_grad_(True)
            dY = torch.randn(T, H, device=dev, dtype=bf)

            def fb():
                Xg.grad = None
                rms.Fast_RMS_Layernorm.apply(Xg, W, 1e-5, False).backward(dY)
            t_fb = timeit(fb)
            emit("rms_fwd+bwd", t_fb, 5 * T * H * 2 + 2 * H * 2 + 2 * T * 4, T=T)
        except Exception as e:
            print("rms failed", repr(e))
        try:
            Q = torch.randn(1, T, Hq, D, device=dev, dtype=bf).transpose(1, 2)
            Kk = torch.randn(1, T, Hk, D, device=dev, dtype=bf).transpose(1, 2)
            cos = torch.randn(T, D, device=dev, dtype=bf)
            sin = torch.randn(T, D, device=dev, dtype=bf)
            idx = torch.arange(T, device=dev, dtype=torch.int32)
            # the reference clones non-contiguous Q/K (rope_embedding.py:293-294): that copy is part of its cost
            emit("rope_qk (strided views, as the model calls it)", timeit(lambda: rope.fast_rope_embedding(Q, Kk, cos, sin, idx)),
                 2 * T * (Hq + Hk) * D * 2 + 2 * T * (D // 2) * 2, T=T)
        except Exception as e:
            print("rope failed", repr(e))
        try:
            e_ = torch.randn(T, I, device=dev, dtype=bf)
            g_ = torch.randn(T, I, device=dev, dtype=bf)
            DW = torch.randn(T, I, device=dev, dtype=bf)
            emit("swiglu_fwd", timeit(lambda: sw.swiglu_fg_kernel(e_.view(1, T, I), g_.view(1, T, I))), 3 * T * I * 2, T=T)
            emit("swiglu_bwd", timeit(lambda: sw.swiglu_DWf_DW_dfg_kernel(DW, e_, g_)), 6 * T * I * 2, T=T)
        except Exception as e:
            print("swiglu failed", repr(e))
        try:
            rows = min(T, 4096)
            logits = torch.randn(1, rows, V, device=dev, dtype=bf)
            labels = torch.randint(0, V, (1, rows), device=dev)
            emit("ce_fwd (4096-row chunk)", timeit(lambda: ce.Fast_CrossEntropyLoss.apply(logits.view(rows, V), labels.view(-1), 0, 0), iters=5, warmup=2),
                 rows * V * 2 + rows * 16, T=rows)
            lg = logits.view(rows, V).clone().requires_grad_(True)

            def cefb():
                lg.grad = None
                ce.Fast_CrossEntropyLoss.apply(lg * 1.0, labels.view(-1), 0, 0).sum().backward()
            emit("ce_fwd+bwd (+1 copy for the leaf guard)", timeit(cefb, iters=5, warmup=2), 3 * rows * V * 2, T=rows)
            del logits, lg
        except Exception as e:
            print("ce failed", repr(e))
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
