"""Time torch SDPA (causal, GQA 32:8, D=128, bf16) forward and backward with each ROCm flash-attention library the
wheel offers (AOTriton default, CK if built in)."""
import json
import sys

import torch
import torch.nn.functional as F

B, Hq, Hk, T, D = 4, 32, 8, 2048, 128
dev = "cuda"


def bench(tag):
    q = torch.randn(B, Hq, T, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, Hk, T, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, Hk, T, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
    for gqa in (True, False):
        kk, vv = (k, v) if gqa else (k.repeat_interleave(4, 1), v.repeat_interleave(4, 1))
        try:
            for _ in range(3):
                o = F.scaled_dot_product_attention(q, kk, vv, is_causal=True, enable_gqa=gqa)
                o.backward(torch.ones_like(o))
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            n = 10
            tf = tb = 0.0
            for _ in range(n):
                ev[0].record()
                o = F.scaled_dot_product_attention(q, kk, vv, is_causal=True, enable_gqa=gqa)
                ev[1].record()
                o.backward(torch.ones_like(o))
                ev[2].record()
                torch.cuda.synchronize()
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
            fl = 4.0 * B * Hq * T * T * D / 2
            print(json.dumps(dict(lib=tag, gqa=gqa, fwd_ms=round(tf / n, 3), bwd_ms=round(tb / n, 3),
                                  fwd_TF=round(fl / (tf / n * 1e-3) / 1e12, 1),
                                  bwd_TF=round(2.5 * fl / (tb / n * 1e-3) / 1e12, 1))), flush=True)
        except Exception as e:
            print(json.dumps(dict(lib=tag, gqa=gqa, error=str(e)[:300])), flush=True)


bench("default:" + str(torch.backends.cuda.preferred_rocm_fa_library()))
for lib in ("ck", "aotriton"):
    try:
        torch.backends.cuda.preferred_rocm_fa_library(lib)
        bench(lib + ":" + str(torch.backends.cuda.preferred_rocm_fa_library()))
    except Exception as e:
        print(json.dumps(dict(lib=lib, error=str(e)[:300])), flush=True)
