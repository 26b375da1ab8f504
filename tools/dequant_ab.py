#!/usr/bin/env python
"""A/B of the row-major NF4 dequant kernels (UAMD_TUNE_DEQUANT_X4 = knob 8: 0 = one 8-element group per lane, 1 = four
groups per lane per trip) at the step's weight shapes. Each shape rotates over enough distinct weights AND distinct output
buffers that nothing is served from the 256 MB Infinity Cache; absmax pre-dequantised to fp32 as in the training step.
One JSON line per (shape, variant): us per launch, TB/s against the algorithmic 2.516 B/param."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib  # noqa: E402
from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    for rows, cols in ((4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336)):
        n = rows * cols
        copies = max(4, int(1.2e9 // (2 * n)))
        ws = []
        for _ in range(copies):
            W = (torch.randn(rows, cols, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
            packed, qs = quantize_nf4(W, compress_statistics=True)
            dequantize_nf4(packed, qs, cache_absmax=True)        # builds the fp32 absmax cache
            ws.append((packed, qs, torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)))
            del W
        res = {}
        for rnd in range(3):
            for knob in (0, 1):
                assert L.uamd_set_tuning(8, knob) == 0
                for p, q, o in ws:
                    dequantize_nf4(p, q, out=o)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    for p, q, o in ws:
                        dequantize_nf4(p, q, out=o)
                e.record()
                torch.cuda.synchronize()
                res.setdefault(knob, []).append(s.elapsed_time(e) * 1e3 / (4 * copies))
        L.uamd_set_tuning(8, 1)
        ok = True
        for p, q, o in ws[:2]:
            L.uamd_set_tuning(8, 0)
            a = dequantize_nf4(p, q, cache_absmax=True).clone()
            L.uamd_set_tuning(8, 1)
            ok = ok and bool(torch.equal(a, dequantize_nf4(p, q, cache_absmax=True)))
        for knob in (0, 1):
            us = sorted(res[knob])[1]
            print(json.dumps({"shape": [rows, cols], "variant": "x4" if knob else "one_group", "us": round(us, 2),
                              "TBps": round(2.516 * n / us / 1e6, 3), "of_8TBps": round(2.516 * n / us / 1e6 / 8, 3),
                              "copies": copies, "bit_identical": ok}), flush=True)
        del ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
