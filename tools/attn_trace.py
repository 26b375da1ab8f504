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


def block_timeline(tb, nblocks):
    """Every CU's sequence of blocks from the per-wave stamps (entry, loop start, loop end, stores issued, stores done, HW_ID,
    XCC_ID, tiles): where the time of a block goes outside its tile loop, and the gap between two blocks on one CU."""
    tb = tb[:nblocks]
    start = tb[:, :, 0].min(1)
    loop0, loop1 = tb[:, :, 1].max(1), tb[:, :, 2].max(1)
    issued, done = tb[:, :, 3].max(1), tb[:, :, 4].max(1)
    hw, xcc, nt = tb[:, 0, 5], tb[:, 0, 6] & 0xf, tb[:, 0, 7]
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)       # (xcc, se, sh, cu)
    d = lambda a, b_: (a - b_) & 0xffffffff                                                          # noqa: E731
    pro, loop, epi, tail = d(loop0, start), d(loop1, loop0), d(issued, loop1), d(done, issued)
    print(f"block timeline: {nblocks} blocks on {len(set(cu.tolist()))} distinct CUs (by XCC / SE / SH / CU id)")
    print(f"  prologue (entry -> tile loop)        mean {pro.mean():8.0f}  p50 {np.median(pro):8.0f}  p90 {np.percentile(pro, 90):8.0f} cycles")
    print(f"  tile loop per tile                   mean {(loop / np.maximum(nt, 1)).mean():8.0f} cycles  ({nt.mean():.1f} tiles per block)")
    print(f"  epilogue (loop end -> stores issued) mean {epi.mean():8.0f}  p90 {np.percentile(epi, 90):8.0f} cycles")
    print(f"  store tail (issued -> vmcnt(0))      mean {tail.mean():8.0f}  p90 {np.percentile(tail, 90):8.0f} cycles")
    gaps, firsts, spans = [], [], []
    t0 = int(start.min())
    for c in sorted(set(cu.tolist())):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(d(start[idx], t0))]
        firsts.append(int(d(start[idx[0]], t0)))
        spans.append(int(d(done[idx[-1]], t0)))
        for a_, b_ in zip(idx[:-1], idx[1:]):
            gaps.append(int(d(start[b_], done[a_])))
    gaps = np.array(gaps) if gaps else np.zeros(1)
    gaps = np.where(gaps > (1 << 31), gaps - (1 << 32), gaps)
    print(f"  gap between two blocks on one CU (stores done -> next entry): mean {gaps.mean():8.0f}  p50 {np.median(gaps):8.0f}  "
          f"p90 {np.percentile(gaps, 90):8.0f} cycles over {len(gaps)} hand-overs")
    print(f"  first entry per CU after the first of all: mean {np.mean(firsts):8.0f}  max {np.max(firsts):8.0f} cycles")
    print(f"  last block done per CU:  min {np.min(spans):8.0f}  mean {np.mean(spans):8.0f}  max {np.max(spans):8.0f} cycles (= the kernel)")
    per_cu = [len(np.where(cu == c)[0]) for c in sorted(set(cu.tolist()))]
    print(f"  blocks per CU: min {min(per_cu)} max {max(per_cu)}")


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
    trace = torch.zeros(256 * 8 * 16 + 4096 * 8 * 8, device="cuda", dtype=torch.int32)
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
    full = trace.cpu().numpy().astype("int64") & 0xffffffff
    t = full[:256 * 8 * 16].reshape(256, 8, 16)
    if os.environ.get("UAMD_ATTN_VAR", "0") in ("0", ""):
        # ping-pong kernel: 8 stamps per tile: start P1 | end P1 | after barrier | end P2 | after barrier | end P3 work |
        # after vmcnt + barrier | end P4; stamp 0 of the next tile closes the last barrier
        names = ["P1: 16 K row reads issued", "P1: wait for the reads", "barrier", "P2: 16 S MFMAs + 4 DMA pieces", "barrier",
                 "P3: 32 V^T reads + mask + max leaves", "P3: max tree, rescale test, P^T step 0", "P3: wait for the reads",
                 "P3: vmcnt (next tile landed)", "barrier", "P4: 16 PV MFMAs + P^T steps 1-3", "barrier (to the next tile's P1)"]
        NS = 12
        for grp, ws in (("leading waves 0-3", range(0, 4)), ("trailing waves 4-7", range(4, 8))):
            acc, n = np.zeros(NS), 0
            for blk in range(256):
                for w in ws:
                    ts = [int(x) for x in t[blk, w][:NS + 1]]
                    if 0 in ts:
                        continue
                    acc += [(ts[i + 1] - ts[i]) & 0xffffffff for i in range(NS)]
                    n += 1
            print(f"{grp}: {n} (block, wave) samples of tile 8")
            for i in range(NS):
                print(f"  {names[i]:44s} {acc[i] / max(n, 1):8.0f} cycles")
            print(f"  {'tile period':44s} {acc.sum() / max(n, 1):8.0f} cycles")
        block_timeline(full[32768:].reshape(4096, 8, 8), B * Hk * ((T + 63) // 64))
        return
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
