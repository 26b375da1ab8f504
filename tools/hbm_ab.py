"""A/B of the streaming (HBM-bound) kernels on one MI355X at the bench's per-launch sizes (T = 8192 tokens):
non-temporal load/store modes for RMSNorm / SwiGLU, the two transposing NF4 dequant kernels, lora_xa, lora_tn.
Interleaved rounds, min over rounds; GB/s = ALGORITHMIC bytes / time."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
import unsloth_amd.kernels as K  # noqa: E402
from unsloth_amd.kernels import utils as U  # noqa: E402
from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4  # noqa: E402

DEV = "cuda"
L = _lib.lib()


def run(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def ab(out, name, nbytes, cands, rounds=5):
    only = os.environ.get("HBM_AB_ONLY")          # substring filter on the kernel label
    if only and only not in name:
        return
    for f in cands.values():
        f()
    torch.cuda.synchronize()
    best = {k: 1e9 for k in cands}
    for _ in range(rounds):
        for k, f in cands.items():
            best[k] = min(best[k], run(f))
    rec = dict(kernel=name, MB=round(nbytes / 1e6, 1), **{k: dict(us=round(v * 1e6, 1), GBps=round(nbytes / v / 1e9))
                                                         for k, v in best.items()})
    print(json.dumps(rec), flush=True)
    if out:
        out.write(json.dumps(rec) + "\n")
        out.flush()


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    only = os.environ.get("HBM_AB_ONLY")          # substring filter on the kernel label
    bf = torch.bfloat16
    T, H, I = 8192, 4096, 14336
    X = torch.randn(T, H, device=DEV, dtype=bf)
    W = torch.rand(H, device=DEV, dtype=bf)
    dY = torch.randn(T, H, device=DEV, dtype=bf)
    r = torch.rand(T, device=DEV)

    def nt(mode, f):
        def g():
            L.uamd_set_tuning(2, mode)
            return f()
        return g

    def rms_b():
        L.uamd_rms_layernorm_bwd(_lib.ptr(dY), _lib.ptr(dY), _lib.ptr(X), _lib.ptr(W), _lib.ptr(r), T, H, H, H, H, 0, 2, 2,
                                 _lib.stream_of(X))
    def var(knob, v, f, m=0):
        def g():
            L.uamd_set_tuning(knob, v)
            L.uamd_set_tuning(2, m)
            return f()
        return g

    Rs = torch.randn(T, H, device=DEV, dtype=bf)
    Hb = torch.empty(T, H, device=DEV, dtype=bf)
    Yb = torch.empty(T, H, device=DEV, dtype=bf)

    def add_f():
        L.uamd_add_rms_layernorm_fwd(_lib.ptr(X), _lib.ptr(Rs), _lib.ptr(W), _lib.ptr(Hb), _lib.ptr(Yb), _lib.ptr(r), T, H,
                                     H, H, H, H, 1e-5, 2, 2, _lib.stream_of(X))

    def add_b():
        L.uamd_add_rms_layernorm_bwd(_lib.ptr(dY), _lib.ptr(Rs), _lib.ptr(dY), _lib.ptr(X), _lib.ptr(W), _lib.ptr(r), T, H,
                                     H, H, H, H, 2, 2, _lib.stream_of(X))
    rf = lambda: K.Fast_RMS_Layernorm.apply(X, W, 1e-5, False)
    for T_ in (T,):
        ab(out, "rms_fwd", 2 * T * H * 2 + H * 2 + T * 4,
           {"wave": var(5, 0, rf), "wave_nt1": var(5, 0, rf, 1), "rowblock": var(5, 1, rf), "rowblock_nt1": var(5, 1, rf, 1),
            "rowblock_nt3": var(5, 1, rf, 3)})
        ab(out, "rms_bwd", 3 * T * H * 2 + H * 2 + T * 4,
           {"wave": var(5, 0, rms_b), "wave_nt1": var(5, 0, rms_b, 1), "rowblock": var(5, 1, rms_b),
            "rowblock_nt1": var(5, 1, rms_b, 1), "rowblock_nt3": var(5, 1, rms_b, 3)})
        ab(out, "add_rms_fwd", 4 * T * H * 2 + H * 2 + T * 4,
           {"wave": var(5, 0, add_f), "rowblock": var(5, 1, add_f), "rowblock_nt1": var(5, 1, add_f, 1)})
        ab(out, "add_rms_bwd", 4 * T * H * 2 + H * 2 + T * 4,
           {"wave": var(5, 0, add_b), "rowblock": var(5, 1, add_b), "rowblock_nt1": var(5, 1, add_b, 1)})
    L.uamd_set_tuning(5, 0)
    L.uamd_set_tuning(2, 0)
    e = torch.randn(T, I, device=DEV, dtype=bf)
    g_ = torch.randn(T, I, device=DEV, dtype=bf)
    DW = torch.randn(T, I, device=DEV, dtype=bf)
    sf = lambda: K.swiglu_fg_kernel(e, g_)
    sb = lambda: K.swiglu_DWf_DW_dfg_kernel(DW, e, g_)
    # note: the forward launcher flips bit0 of the nt mode (default = non-temporal loads)
    ab(out, "swiglu_fwd", 3 * T * I * 2, {f"v{v}_nt{m}": var(0, v, sf, m) for v in (0, 1, 2) for m in (0, 1, 2)})
    ab(out, "swiglu_bwd", 6 * T * I * 2, {f"v{v}_nt{m}": var(0, v, sb, m) for v in (0, 1, 2) for m in (0, 1, 3)})
    L.uamd_set_tuning(0, 0)
    L.uamd_set_tuning(2, 0)
    # transposing dequant
    Wd = (torch.randn(I, H, device=DEV) * 0.02).to(bf)
    packed, qs = quantize_nf4(Wd)

    def dq(knob):
        def f():
            L.uamd_set_tuning(3, knob)
            return dequantize_nf4(packed, qs, transpose=True, use_global_buffer=True)
        return f
    def dq_pad(pk, q, pad):
        rows, cols = q.shape
        buf = torch.empty(cols, rows + pad, device=DEV, dtype=bf)

        def f():
            L.uamd_set_tuning(3, 1)
            return dequantize_nf4(pk, q, out=buf[:, :rows], transpose=True)
        return f
    ab(out, "nf4_dequant_T gate", I * H * 2.516, {"t256": dq(1), "t256_rowfast": dq(2),
                                                   "plain": lambda: dequantize_nf4(packed, qs, use_global_buffer=True)})
    Wd2 = (torch.randn(H, I, device=DEV) * 0.02).to(bf)
    packed2, qs2 = quantize_nf4(Wd2)

    def dq2(knob):
        def f():
            L.uamd_set_tuning(3, knob)
            return dequantize_nf4(packed2, qs2, transpose=True, use_global_buffer=True)
        return f
    ab(out, "nf4_dequant_T down", I * H * 2.516, {"t256": dq2(1), "t256_rowfast": dq2(2), 
                                                   "plain": lambda: dequantize_nf4(packed2, qs2, use_global_buffer=True)})
    Wd3 = (torch.randn(H, H, device=DEV) * 0.02).to(bf)
    packed3, qs3 = quantize_nf4(Wd3)

    def dq3(knob):
        def f():
            L.uamd_set_tuning(3, knob)
            return dequantize_nf4(packed3, qs3, transpose=True, use_global_buffer=True)
        return f
    ab(out, "nf4_dequant_T o", H * H * 2.516, {"t256": dq3(1), "t256_rowfast": dq3(2), 
                                               "plain": lambda: dequantize_nf4(packed3, qs3, use_global_buffer=True)})
    L.uamd_set_tuning(3, 1)
    # lora_xa / lora_tn
    A3 = [torch.nn.Parameter(torch.randn(16, H, device=DEV) * 0.02) for _ in range(3)]
    def xav(v, f):
        def g():
            U.LORA_XA_V2 = (v == 2)
            return f()
        return g
    ab(out, "lora_xa qkv (R=48)", T * H * 2, {"v1": xav(1, lambda: U.lora_xa(X, A3)), "v2": xav(2, lambda: U.lora_xa(X, A3))})
    ab(out, "lora_xa o (R=16)", T * H * 2, {"v1": xav(1, lambda: U.lora_xa(X, A3[:1])), "v2": xav(2, lambda: U.lora_xa(X, A3[:1]))})
    A1 = [torch.nn.Parameter(torch.randn(16, I, device=DEV) * 0.02)]
    ab(out, "lora_xa down (R=16,K=14336)", T * I * 2, {"v1": xav(1, lambda: U.lora_xa(e, A1)), "v2": xav(2, lambda: U.lora_xa(e, A1))})
    U.LORA_XA_V2 = True
    P = torch.randn(T, 16, device=DEV)
    ab(out, "lora_tn dA (Z=X)", T * H * 2, {"v1": lambda: U.lora_tn([(P, X, 16, False, 1.0)])})
    ab(out, "lora_tn dB (Z=e)", T * I * 2, {"v1": lambda: U.lora_tn([(P, e, 16, True, 1.0)])})
    ab(out, "lora_tn MLP (6 problems)", (3 * T * I + 3 * T * H) * 2,
       {"v1": lambda: U.lora_tn([(P, e, 16, False, 1.0), (P, dY, 16, True, 1.0), (P, X, 16, False, 1.0),
                                 (P, g_, 16, True, 1.0), (P, X, 16, False, 1.0), (P, DW, 16, True, 1.0)])})


if __name__ == "__main__":
    main()
