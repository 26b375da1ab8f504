"""Per-kernel micro-benchmark on one MI355X: achieved HBM GB/s (or TFLOP/s) per launch against the
algorithmic bytes/flops of SURVEY 8(d). Timing: torch.cuda.Event pairs on the current stream (the
stream the C ABI launches on). Writes one JSON object per line to stdout / --out."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unsloth_amd.kernels as K  # noqa: E402
from unsloth_amd.kernels import utils as U  # noqa: E402
from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4  # noqa: E402

DEV = "cuda"
HBM_PEAK = 8000.0      # GB/s, spec (MI355X_MICROARCH.md)
MFMA_PEAK = 2500.0     # TFLOP/s dense bf16


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


def emit(out, name, secs, nbytes=None, flops=None, **kw):
    rec = dict(kernel=name, us=round(secs * 1e6, 2), **kw)
    if nbytes is not None:
        rec.update(GBps=round(nbytes / secs / 1e9, 1), frac_hbm=round(nbytes / secs / 1e9 / HBM_PEAK, 3))
    if flops is not None:
        rec.update(TFLOPs=round(flops / secs / 1e12, 1), frac_mfma=round(flops / secs / 1e12 / MFMA_PEAK, 3))
    print(json.dumps(rec), flush=True)
    if out:
        out.write(json.dumps(rec) + "\n")
        out.flush()


def main():
    if os.environ.get('UAMD_DBG_LIB'):
        from unsloth_amd import _lib as _l
        _l.LIB_PATH = os.environ['UAMD_DBG_LIB']
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--tokens", type=int, nargs="+", default=[2048, 8192])
    ap.add_argument("--skip-gemm", action="store_true")
    ap.add_argument("--only-gemm", action="store_true")
    ap.add_argument("--gemm-tokens", type=int, nargs="+", default=None)
    a = ap.parse_args()
    out = open(a.out, "w") if a.out else None
    bf = torch.bfloat16
    H, I, V, Hq, Hk, D = 4096, 14336, 128256, 32, 8, 128
    for T in ([] if a.only_gemm else a.tokens):
        X = torch.randn(T, H, device=DEV, dtype=bf)
        W = torch.rand(H, device=DEV, dtype=bf)
        dY = torch.randn(T, H, device=DEV, dtype=bf)
        Y = K.Fast_RMS_Layernorm.apply(X, W, 1e-5, False)
        emit(out, "rms_fwd", timeit(lambda: K.Fast_RMS_Layernorm.apply(X, W, 1e-5, False)), 2 * T * H * 2 + H * 2 + T * 4, T=T)
        from unsloth_amd import _lib
        r = torch.rand(T, device=DEV)

        def rms_b():
            _lib.lib().uamd_rms_layernorm_bwd(_lib.ptr(dY), _lib.ptr(dY), _lib.ptr(X), _lib.ptr(W), _lib.ptr(r), T, H,
                                              H, H, H, 0, 2, 2, _lib.stream_of(X))
        emit(out, "rms_bwd", timeit(rms_b), 3 * T * H * 2 + H * 2 + T * 4, T=T)
        # rope
        Q = torch.randn(1, T, Hq, D, device=DEV, dtype=bf).transpose(1, 2)
        Kk = torch.randn(1, T, Hk, D, device=DEV, dtype=bf).transpose(1, 2)
        cos = torch.randn(T, D, device=DEV, dtype=bf)
        sin = torch.randn(T, D, device=DEV, dtype=bf)
        idx = torch.arange(T, device=DEV, dtype=torch.int32)
        emit(out, "rope_qk", timeit(lambda: K.fast_rope_embedding(Q, Kk, cos, sin, idx)),
             2 * T * (Hq + Hk) * D * 2 + 2 * T * (D // 2) * 2, T=T)
        # swiglu
        e = torch.randn(T, I, device=DEV, dtype=bf)
        g = torch.randn(T, I, device=DEV, dtype=bf)
        DW = torch.randn(T, I, device=DEV, dtype=bf)
        emit(out, "swiglu_fwd", timeit(lambda: K.swiglu_fg_kernel(e, g)), 3 * T * I * 2, T=T)
        emit(out, "swiglu_bwd", timeit(lambda: K.swiglu_DWf_DW_dfg_kernel(DW, e, g)), 6 * T * I * 2, T=T)
        del e, g, DW
        # CE
        if T <= 4096:
            logits = torch.randn(T, V, device=DEV, dtype=bf)
            labels = torch.randint(0, V, (T,), device=DEV)
            from unsloth_amd.kernels.cross_entropy_loss import _ce_forward, _ce_backward_
            losses, lse = _ce_forward(logits, labels, 0, 0)
            emit(out, "ce_fwd", timeit(lambda: _ce_forward(logits, labels, 0, 0)), T * V * 2 + T * 16, T=T)
            dl = torch.ones(T, device=DEV)
            emit(out, "ce_bwd", timeit(lambda: _ce_backward_(logits, dl, lse, labels, 0, 0)), 2 * T * V * 2, T=T)
            del logits
    # NF4 dequant
    Wd = (torch.randn(I, H, device=DEV) * 0.02).to(bf)
    packed, qs = quantize_nf4(Wd)
    nparam = I * H
    emit(out, "nf4_dequant", timeit(lambda: dequantize_nf4(packed, qs, use_global_buffer=True)), nparam * 2.516, params=nparam)
    emit(out, "nf4_dequant_T", timeit(lambda: dequantize_nf4(packed, qs, transpose=True, use_global_buffer=True)), nparam * 2.516, params=nparam)
    # fused residual add + norm
    T = a.tokens[-1]
    from unsloth_amd.kernels.rms_layernorm import Fast_Add_RMS_Layernorm
    X = torch.randn(T, H, device=DEV, dtype=bf)
    Rr = torch.randn(T, H, device=DEV, dtype=bf)
    Wn = torch.rand(H, device=DEV, dtype=bf)
    emit(out, "add_rms_fwd", timeit(lambda: Fast_Add_RMS_Layernorm.apply(X, Rr, Wn, 1e-5)), 4 * T * H * 2 + H * 2 + T * 4, T=T)
    # attention (causal GQA 32:8, d 128), B x 2048
    from unsloth_amd.kernels import attention as A_
    Bq, S = T // 2048, 2048
    qkv = torch.randn(Bq, S, (Hq + 2 * Hk) * D, device=DEV, dtype=bf)
    q = qkv[..., :Hq * D].view(Bq, S, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(Bq, S, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(Bq, S, Hk, D)
    o, lse = A_.attn_forward(q, k, v)
    do = torch.randn_like(o)
    fl = 4.0 * Bq * Hq * S * S * D / 2
    emit(out, "attn_fwd", timeit(lambda: A_.attn_forward(q, k, v)), flops=fl, T=T)
    emit(out, "attn_bwd(dq+dkdv)", timeit(lambda: A_.attn_backward(do, q, k, v, o, lse)), flops=2.5 * fl, T=T)
    # LoRA side products
    A3 = [torch.nn.Parameter(torch.randn(16, H, device=DEV) * 0.02) for _ in range(3)]
    emit(out, "lora_xa2 q|k|v (R=48)", timeit(lambda: U.lora_xa(X, A3)), T * H * 2, T=T)
    emit(out, "lora_xa2 (R=16, K=4096)", timeit(lambda: U.lora_xa(X, A3[:1])), T * H * 2, T=T)
    Xi = torch.randn(T, I, device=DEV, dtype=bf)
    Ad = [torch.nn.Parameter(torch.randn(16, I, device=DEV) * 0.02)]
    emit(out, "lora_xa2 (R=16, K=14336)", timeit(lambda: U.lora_xa(Xi, Ad)), T * I * 2, T=T)
    P = torch.randn(T, 16, device=DEV)
    emit(out, "lora_tn 6 problems (MLP block)", timeit(lambda: U.lora_tn([(P, Xi, 16, False, 1.0), (P, X, 16, True, 1.0),
                                                                          (P, X, 16, False, 1.0), (P, Xi, 16, True, 1.0),
                                                                          (P, X, 16, False, 1.0), (P, Xi, 16, True, 1.0)])),
         (3 * T * I + 3 * T * H) * 2, T=T)
    del Xi
    if a.skip_gemm:
        return
    # GEMMs: the step's shapes on the shipped kernel, hipBLASLt beside it
    for T in (a.gemm_tokens or a.tokens[-1:]):
        X = torch.randn(T, H, device=DEV, dtype=bf)
        for (N, Kd, tag) in ((H, H, "o_proj"), (I, H, "gate_proj"), (H, I, "down_proj"), (H, H + 2 * Hk * D, "qkv_dx_merged")):
            Xin = X if Kd == H else torch.randn(T, Kd, device=DEV, dtype=bf)
            Wf = (torch.randn(N, Kd, device=DEV) * 0.02).to(bf)
            fl = 2.0 * T * N * Kd
            emit(out, f"torch_matmul_{tag}", timeit(lambda: Xin @ Wf.t()), flops=fl, T=T)
            emit(out, f"gemm_nt_{tag}", timeit(lambda: U.lora_linear_forward(Xin, [(Wf, None, None, None, None)])), flops=fl, T=T,
                 kernel_used=U._launch_gemm.__name__ if False else ("256" if U._use_gemm256(T, Kd, [N]) else "128"))
            Aa = torch.nn.Parameter(torch.randn(16, Kd, device=DEV) * 0.02)
            Bb = torch.nn.Parameter(torch.randn(N, 16, device=DEV) * 0.02)
            t_all = timeit(lambda: U.lora_linear_forward(Xin, [(Wf, None, Aa, Bb, 1.0)]))
            emit(out, f"gemm_nt+lora_{tag} (X A^T launch + GEMM with the rank block)", t_all, flops=fl, T=T)
            use256 = U._use_gemm256(T, Kd, [N])
            t_xa = timeit(lambda: U._xa_and_rank_block(Xin, [Aa], use256))
            emit(out, f"gemm_nt+lora_{tag} (GEMM alone = above minus the X A^T launch)", max(t_all - t_xa, 1e-9), flops=fl, T=T,
                 xa_us=round(t_xa * 1e6, 2))
            del Wf


if __name__ == "__main__":
    main()
