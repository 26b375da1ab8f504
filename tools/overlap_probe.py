"""Does an HBM-bound side kernel hide under an attention kernel on a second stream?  (what a next round could do with
the NF4 decodes -- 6.2 ms of a 222 ms step -- that have no data dependence on the layer before them.)
For each attention kernel (forward one-block / persistent, backward pair) at 4 x 2048 tokens: the kernel alone, an NF4 decode of
gate|up-sized weights alone, both back to back on one stream, both on two streams (fork / join by events). One JSON line each."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.kernels import attention as A  # noqa: E402
from unsloth_amd.nf4 import dequantize_nf4, quantize_nf4  # noqa: E402

dev, bf = "cuda", torch.bfloat16
L = _lib.lib()
B, T, Hq, Hk, D = 4, 2048, 32, 8, 128
torch.manual_seed(0)
qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device=dev, dtype=bf)
q = qkv[..., :Hq * D].view(B, T, Hq, D)
k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
o, lse = A.attn_forward(q, k, v)
do = torch.randn_like(o)
W = torch.randn(14336, 4096, device=dev, dtype=bf) * 0.02
packed, qs = quantize_nf4(W)
out = torch.empty_like(W)
side = torch.cuda.Stream()


def dec():
    dequantize_nf4(packed, qs, out=out)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n * 1e3)
    return round(sorted(ts)[2], 1)


def two_streams(main_fn):
    def run():
        fork = torch.cuda.Event()
        fork.record()
        with torch.cuda.stream(side):
            side.wait_event(fork)
            dec()
            join = torch.cuda.Event()
            join.record()
        main_fn()
        torch.cuda.current_stream().wait_event(join)
    return run


def fwd1():
    L.uamd_set_tuning(4, 1)
    A.attn_forward(q, k, v)
    L.uamd_set_tuning(4, 0)


cases = {"forward, one block per item (96 KiB LDS, 2 waves per SIMD)": fwd1,
         "forward, persistent (160 KiB LDS)": lambda: A.attn_forward(q, k, v),
         "backward pair (dQ: 96 KiB LDS, 2 waves per SIMD; dK/dV: 146 KiB, every register)":
             lambda: A.attn_backward(do, q, k, v, o, lse)}
t_dec = timed(dec)
for name, fn in cases.items():
    rec = dict(attention=name, attention_alone_us=timed(fn), decode_alone_us=t_dec,
               one_stream_us=timed(lambda: (fn(), dec())), two_streams_us=timed(two_streams(fn)))
    rec["hidden_us"] = round(rec["one_stream_us"] - rec["two_streams_us"], 1)
    print(json.dumps(rec), flush=True)
