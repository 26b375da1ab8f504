"""Phase timeline of the attention forward kernel (attention.hip built with -DUAMD_ATTN_TRACE): s_memtime stamps
around [vmcnt wait | barrier | DMA issue | S MFMAs | softmax | PV MFMAs] of tiles 8 and 9, per wave.
usage (GPU box): python tools/attn_trace.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    so = "/tmp/uamd_libattntrace.so"
    srcs = [os.path.join(ROOT, "unsloth_amd/csrc", f) for f in ("attention.hip", "abi.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-mcode-object-version=5", "-ffp-contract=off", "-DUAMD_ATTN_TRACE",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "unsloth_amd/csrc"),
                           *srcs, "-o", so])
    L = ctypes.CDLL(so)
    B, T, Hq, Hk, D = 4, 2048, 32, 8, 128
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, device="cuda", dtype=torch.bfloat16)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    o = torch.empty(B, T, Hq, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, Hq, T, device="cuda", dtype=torch.float32)
    trace = torch.zeros(256 * 8 * 16, device="cuda", dtype=torch.int32)
    L.uamd_debug_attn_trace.argtypes = [ctypes.c_void_p]
    assert L.uamd_debug_attn_trace(trace.data_ptr()) == 0
    st = (ctypes.c_int64 * 12)(*[x for t_ in (q, k, v, o) for x in (t_.stride(0), t_.stride(1), t_.stride(2))])
    fn = L.uamd_attn_fwd
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_int64)] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int,
                                                                                                 ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    for _ in range(3):
        rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), st, B, T, Hq, Hk, D, T,
                1.0 / D ** 0.5, 1, None, 1, None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    t = trace.view(256, 8, 16).cpu().numpy().astype("int64") & 0xffffffff
    names = ["vmcnt wait", "barrier", "DMA issue", "(skip test)", "S MFMAs", "softmax", "PV MFMAs"]
    acc = np.zeros(7)
    n = 0
    for blk in range(256):          # the first 256 blocks are the heaviest q tiles (32 kv tiles each)
        for w in range(8):
            for tile in (0, 1):
                ts = t[blk, w, tile * 8: tile * 8 + 7]
                if ts[0] == 0 or ts[6] == 0:
                    continue
                d = [(int(ts[i + 1]) - int(ts[i])) & 0xffffffff for i in range(6)]
                acc[:6] += d
                n += 1
            a, b_ = t[blk, w, 0], t[blk, w, 8]
            if a and b_:
                acc[6] += (int(b_) - int(a)) & 0xffffffff
    print(f"waves x tiles sampled: {n}")
    for i in range(6):
        print(f"  {names[i] if i < 3 else names[i + 1]:12s} {acc[i] / n:8.0f} cycles")
    print(f"  tile period (stamp 0 of tile 8 -> stamp 0 of tile 9): {acc[6] / (n / 2):8.0f} cycles")


if __name__ == "__main__":
    main()
