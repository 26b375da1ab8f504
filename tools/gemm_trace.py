"""Slot-level timeline of the 256x256 ping-pong GEMM (gemm256.hip built with -DUAMD_G256_TRACE into
gpurun_out/libtrace.so): s_memtime stamps at every barrier of K tiles 16 and 17, per wave.
usage (GPU box): python tools/gemm_trace.py [slots=8|4]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    alias = len(sys.argv) > 2 and sys.argv[2] == "alias"     # lda = ldb = 0: every row is row 0 (L1-resident operands)
    so = "/tmp/uamd_libtrace.so"
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        srcs = [os.path.join(ROOT, "unsloth_amd/csrc", f) for f in ("gemm256.hip", "abi.hip")]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-mcode-object-version=5", "-ffp-contract=off", "-DUAMD_G256_TRACE=" + os.environ.get("TRACE_LEVEL", "1"),
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "unsloth_amd/csrc"),
                               *srcs, "-o", so])
    L = ctypes.CDLL(so)
    from unsloth_amd import _lib
    G = _lib.GemmGroup
    M, N, K = 8192, 14336, 4096
    X = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    trace = torch.zeros(1024 * 8 * 32, device="cuda", dtype=torch.int32)
    L.uamd_debug_g256_trace.argtypes = [ctypes.c_void_p]
    assert L.uamd_debug_g256_trace(trace.data_ptr()) == 0
    g = G()
    g.B, g.ldb, g.C, g.ldc, g.N = W.data_ptr(), (0 if alias else K), C.data_ptr(), N, N
    g.lora_xa = None
    g.lora_b = None
    fn = L.uamd_gemm_nt_256
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lda = 0 if alias else K
    for _ in range(3):
        rc = fn(X.data_ptr(), lda, M, K, ctypes.byref(g), 1, 0, _lib.dtype_code(torch.bfloat16), None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(10):
        fn(X.data_ptr(), lda, M, K, ctypes.byref(g), 1, 0, _lib.dtype_code(torch.bfloat16), None)
    e_.record()
    torch.cuda.synchronize()
    ms = s_.elapsed_time(e_) / 10
    print(f"wall {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s (instrumented build)")
    P = _lib.lib()
    for al in (0, 1, 0, 1):
        g.ldb = 0 if al else K
        la = 0 if al else K
        for _ in range(3):
            P.uamd_gemm_nt_256(X.data_ptr(), la, M, K, ctypes.byref(g), 1, 0, _lib.dtype_code(torch.bfloat16), None)
        s_.record()
        for _ in range(20):
            P.uamd_gemm_nt_256(X.data_ptr(), la, M, K, ctypes.byref(g), 1, 0, _lib.dtype_code(torch.bfloat16), None)
        e_.record()
        torch.cuda.synchronize()
        ms = s_.elapsed_time(e_) / 20
        print(f"product build, alias={al}: {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s")
    g.ldb = 0 if alias else K
    ref = X @ W.t()
    print("rel err", float((C.float() - ref.float()).norm() / ref.float().norm()))
    t = trace.view(1024, 8, 32).cpu().numpy().astype("int64") & 0xffffffff
    whole = (t[:, :, 31] - t[:, :, 30]) & 0xffffffff
    print(f"whole-tile cycles (main loop of {K // 64} K tiles + epilogue), mean over 1024 blocks x 8 waves: {whole.mean():.0f}"
          f" -> {whole.mean() / (K // 64):.0f} per K tile; wall-implied clock {whole.mean() * (M // 256) * (N // 256) / 256 / (ms * 1e-3) / 1e9:.2f} GHz")
    if os.environ.get("TRACE_LEVEL", "1") != "1":
        return
    nb = slots
    for blk in (0, 5, 300, 777):
        for w in (0, 1, 4, 5):
            ts = t[blk, w]
            # chronological order of stamps inside a tile: [8+i (work done), i (barrier passed)] for i in range(nb)
            seq = []
            for tile in (0, 1):
                for i in range(nb):
                    seq += [ts[tile * 16 + 8 + i], ts[tile * 16 + i]]
            d = [(b - a) & 0xffffffff for a, b in zip(seq[:-1], seq[1:])]
            # d[2i] = wait at barrier i, d[2i+1] = work of the next slot
            print(f"blk {blk:4d} wave {w}: total/tile {((seq[-1] - seq[1]) & 0xffffffff) / (2 - 1.0 / (2*nb)) :.0f}  "
                  + " ".join(f"{x}" for x in d))
    # averages over all waves of all blocks: per-position work and wait
    import numpy as np
    work = np.zeros(2 * nb)
    wait = np.zeros(2 * nb)
    cnt = 0
    for blk in range(0, 1024):
        for w in range(8):
            ts = t[blk, w]
            if ts[0] == 0:
                continue
            seq = []
            for tile in (0, 1):
                for i in range(nb):
                    seq += [ts[tile * 16 + 8 + i], ts[tile * 16 + i]]
            d = np.array([(b - a) & 0xffffffff for a, b in zip(seq[:-1], seq[1:])], dtype=np.float64)
            if w < 4:
                wait[:] += np.append(d[0::2], 0)[: 2 * nb]
                work[:] += np.append(d[1::2], 0)[: 2 * nb]
                cnt += 1
    print("group0 mean barrier-wait per slot:", np.round(wait / cnt).tolist())
    print("group0 mean work per following slot:", np.round(work / cnt).tolist())
    print("mean cycles per K tile (group 0):", round((wait.sum() + work.sum()) / cnt / 2 * (2 * nb) / (2 * nb - 0.5)))


if __name__ == "__main__":
    main()
