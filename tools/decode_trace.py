"""Phase timeline of the decode-step kernels (decode.hip built with -DUAMD_DECODE_TRACE): s_memtime stamps per wave at
[entry | weight loads issued | token staged + table built | barrier | first trip's dot products | t = A x picked up |
reduction | end] of gemv_kernel (weight-row workgroups and the t workgroups apart) and at the phase boundaries of
attn_decode_fused_kernel, for the five launches of one Llama-3-8B decoder-layer step.
usage (GPU box): python tools/decode_trace.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "unsloth_amd", "lib", "libunsloth_amd_dectrace.so")


def build():
    lib = os.path.join(ROOT, "unsloth_amd", "lib")
    obj = "/tmp/decode_trace.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5",
                           "-ffp-contract=off", "-DUAMD_DECODE_TRACE", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "unsloth_amd/csrc"), "-c", os.path.join(ROOT, "unsloth_amd/csrc/decode.hip"),
                           "-o", obj])
    others = [os.path.join(lib, f) for f in sorted(os.listdir(lib)) if f.endswith(".o") and f != "decode.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *others, obj, "-o", SO])


def main():
    build()
    os.environ["UNSLOTH_AMD_LIB"] = SO
    import torch
    from unsloth_amd import _lib
    from unsloth_amd.kernels import decode as D
    from unsloth_amd.nf4 import quantize_nf4
    L = _lib.lib()
    L.uamd_debug_decode_trace.argtypes = [ctypes.c_void_p]
    dev, bf = "cuda", torch.bfloat16
    trace = torch.zeros(1024 * 8 * 16, dtype=torch.int64, device=dev)
    assert L.uamd_debug_decode_trace(trace.data_ptr()) == 0
    H, I, Hq, Hk, Dh = 4096, 14336, 32, 8, 128

    def lora_projs(Ns, K):
        ps = []
        for N in Ns:
            W = (torch.randn(N, K, device=dev) * 0.02).to(bf)
            packed, qs = quantize_nf4(W, compress_statistics=True)
            qs.dtype = bf
            ps.append((packed, qs, torch.nn.Parameter(torch.randn(16, K, device=dev) * 0.02),
                       torch.nn.Parameter(torch.randn(N, 16, device=dev) * 0.02), 1.0, None))
        return ps

    def run(fn, n_tb, waves, names, what):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        trace.zero_()
        fn()
        torch.cuda.synchronize()
        t = trace.view(-1, 16).cpu().numpy().astype(np.int64)
        used = t[:, 0] != 0
        t = t[used]
        blk = np.nonzero(used)[0] // waves
        xcc = t[:, 15]
        # every XCD has its own s_memtime base: a wave's stamps are comparable with its own entry and with waves of its XCD
        t0x = {x: t[xcc == x, 0].min() for x in np.unique(xcc)}
        enter = np.array([t[i, 0] - t0x[xcc[i]] for i in range(len(t))])
        last = np.array([t[i, :13].max() - t0x[xcc[i]] for i in range(len(t))])
        dur_ticks = np.array([t[i, :13].max() - t[i, 0] for i in range(len(t))], dtype=np.float64)
        dur_real = (t[:, 14] - t[:, 13]).astype(np.float64) * 10.0          # ns (s_memrealtime = 100 MHz)
        ok = dur_real > 2000
        if ok.any():
            print(f"   (s_memtime: {np.median(dur_ticks[ok] / dur_real[ok]):.3f} ticks per ns)")
        # chip-wide timeline from s_memrealtime (one 100 MHz counter for all XCDs): when waves entered, when the last one left
        r0 = t[:, 13].min()
        ent_ns = (t[:, 13] - r0) * 10
        end_ns = (t[:, 14] - r0) * 10
        print(f"   wave entry after the launch's first wave: median {int(np.median(ent_ns))} ns, 90 % {int(np.percentile(ent_ns, 90))} ns, "
              f"last {int(ent_ns.max())} ns; last wave leaves at {int(end_ns.max())} ns; a wave lives {int(np.median(end_ns - ent_ns))} ns (median)")
        print(f"\n== {what}: {us:.1f} us per launch back to back (host-bound in eager mode); {len(t)} waves on {len(t0x)} XCDs; "
              f"XCD's first entry -> its last stamp: {int(np.median([last[xcc == x].max() for x in t0x]))} ticks (median over XCDs)")
        for label, sel in (("t workgroups", blk < n_tb), ("weight-row / all workgroups", blk >= n_tb)):
            tt = t[sel]
            if not len(tt):
                continue
            e = enter[sel]
            print(f"  {label} ({len(tt)} waves): entry {int(np.median(e))} [{e.min()} .. {e.max()}] ticks after the XCD's first wave; "
                  f"then ticks since the wave's OWN entry, median [min .. max]")
            for i, nm in enumerate(names):
                if i == 0:
                    continue
                ok = tt[:, i] != 0
                col = tt[ok, i] - tt[ok, 0]
                if len(col):
                    print(f"    {i:2d} {nm:44s} {int(np.median(col)):7d} [{col.min():6d} .. {col.max():6d}]  ({len(col)} waves)")

    gemv_names = ["entry", "weight loads of trip 1 issued", "x staged, tables built", "after the barrier", "t row reduced (t workgroups)",
                  "trip 1 dot products done, poll starts", "t picked up", "trip 1 reduced", "all rows stored", "end",
                  "  (kernarg lines touched)", "  (prologue operand loads issued)", "  (absmax map / NF4 level loads issued)"]
    resid = torch.randn(H, device=dev, dtype=bf)
    delta = torch.randn(H, device=dev, dtype=bf)
    wn = torch.ones(H, device=dev, dtype=bf)
    hbuf = torch.empty(H, device=dev, dtype=bf)
    for name, Ns, K, fused, xin, rt in (
            ("q|k|v (add + RMSNorm in)", (4096, 1024, 1024), H, dict(mode=2, res=resid, norm_w=wn, eps=1e-5, h_out=hbuf), delta, 48),
            ("o", (4096,), H, dict(mode=0), delta, 16),
            ("gate|up (add + RMSNorm in, SwiGLU out)", (I, I), H, dict(mode=2, res=resid, norm_w=wn, eps=1e-5, h_out=hbuf, glu=True), delta, 32),
            ("down", (4096,), I, dict(mode=0), torch.randn(I, device=dev, dtype=bf), 16)):
        projs = lora_projs(Ns, K)
        outbuf = torch.empty(Ns[0] if fused.get("glu") else sum(Ns), device=dev, dtype=bf)
        ks = 1
        while ks * 4096 < K:
            ks *= 2
        run(lambda: D.linear_group(xin, projs, out=outbuf, fused=fused), (rt * ks + 7) // 8, 8, gemv_names, "fused gemv " + name)
        bare = [(p[0], p[1], None, None, None, None) for p in projs]
        run(lambda: D.linear_group(xin, bare, out=outbuf if not fused.get("glu") else None), 0, 8, gemv_names,
            "plain gemv, no adapter, " + name)
    S = 2048
    kc = torch.randn(1, Hk, S, Dh, device=dev, dtype=bf)
    vc = torch.randn(1, Hk, S, Dh, device=dev, dtype=bf)
    qkv = torch.randn(1, (Hq + 2 * Hk) * Dh, device=dev, dtype=bf)
    cos = torch.randn(S, Dh // 2, device=dev, dtype=bf)
    sin = torch.randn(S, Dh // 2, device=dev, dtype=bf)
    kvl = torch.full((1,), S - 1, dtype=torch.int32, device=dev)
    part = torch.empty(1, Hq, S // 128, Dh + 2, dtype=torch.float32, device=dev)
    fpart, cnt = D.fused_attn_workspace(1, Hq, Hk, S, Dh, 128, dev)
    ao = torch.empty(1, Hq * Dh, device=dev, dtype=bf)
    attn_names = ["entry", "q / new k rotated (before the barrier)", "after the barrier", "16 K/V wave-loads issued", "K/V landed",
                  "keys done", "granules published", "-", "-", "own share combined", "end"]
    run(lambda: D.attn_decode_fused(qkv, cos, sin, kvl, kc, vc, ao, fpart, cnt, 128, 0.088, Hq), 0, 4, attn_names,
        "fused attention, context 2048")


if __name__ == "__main__":
    main()
