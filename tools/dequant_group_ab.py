"""NF4 decode of the q | k | v and gate | up groups of Llama-3-8B: one launch per weight against one launch per group
(uamd_nf4_dequantize_multi). One JSON line per group.    python tools/dequant_group_ab.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4, dequantize_nf4_group  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n * 1e3)
    return sorted(ts)[2]


for name, shapes in (("q|k|v", [(4096, 4096), (1024, 4096), (1024, 4096)]), ("gate|up", [(14336, 4096), (14336, 4096)]),
                     ("o", [(4096, 4096)]), ("down", [(4096, 14336)])):
    # several copies so that the packed codes do not sit in the 256 MiB cache between launches
    copies = []
    for c in range(6):
        pks, qss, outs = [], [], []
        for (r, k) in shapes:
            pk, qs = quantize_nf4((torch.randn(r, k, device="cuda") * 0.02).to(torch.bfloat16))
            pks.append(pk)
            qss.append(qs)
            outs.append(torch.empty((r, k), dtype=torch.bfloat16, device="cuda"))
        copies.append((pks, qss, outs))
    it = [0]

    def single():
        pks, qss, outs = copies[it[0] % len(copies)]
        it[0] += 1
        for pk, qs, o in zip(pks, qss, outs):
            dequantize_nf4(pk, qs, out=o)

    def group():
        pks, qss, outs = copies[it[0] % len(copies)]
        it[0] += 1
        dequantize_nf4_group(pks, qss, outs)

    nbytes = sum(r * k for r, k in shapes) * (2 + 0.5 + 4 / 64)
    t1, t2 = timed(single), timed(group)
    print(json.dumps(dict(group=name, single_us=round(t1, 2), grouped_us=round(t2, 2), single_TBps=round(nbytes / t1 / 1e6, 3),
                          grouped_TBps=round(nbytes / t2 / 1e6, 3))), flush=True)
