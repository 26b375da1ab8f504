"""debug: forward attention, 64-rows-per-wave kernel (UAMD_TUNE_ATTN_VAR bit 0) against the 8-wave kernel."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsloth_amd import _lib
if os.environ.get('UAMD_DBG_LIB'):
    _lib.LIB_PATH = os.environ['UAMD_DBG_LIB']
from unsloth_amd.kernels.attention import attn_forward
L = _lib.lib()
for dtype in (torch.bfloat16,):
    for (B, T, Hq, Hk) in ((1, 64, 4, 1), (1, 128, 4, 1), (1, 512, 4, 1)):
        g = torch.Generator().manual_seed(1)
        D = 128
        q = torch.randn(B, T, Hq, D, generator=g).to(dtype).cuda()
        k = torch.randn(B, T, Hk, D, generator=g).to(dtype).cuda()
        v = torch.randn(B, T, Hk, D, generator=g).to(dtype).cuda()
        L.uamd_set_tuning(4, 0)
        o0, l0 = attn_forward(q, k, v, 1 / math.sqrt(D))
        L.uamd_set_tuning(4, 1)
        o1, l1 = attn_forward(q, k, v, 1 / math.sqrt(D))
        torch.cuda.synchronize()
        d = (o1.float() - o0.float())
        bad = ~torch.isfinite(o1.float()) | (d.abs() > 0.05)
        print(dtype, (B, T, Hq, Hk), "lse maxdiff", float((l1 - l0).abs().max()), "o bad frac", float(bad.float().mean()),
              "maxdiff(finite)", float(d[torch.isfinite(d)].abs().max()))
        if bad.any():
            idx = bad.nonzero()
            print("  first bad (b,t,h,d):", idx[:6].tolist(), " bad per d-tile:", [int(bad[..., i * 32:(i + 1) * 32].sum()) for i in range(4)],
                  " bad per t-block of 32:", [int(bad[:, i * 32:(i + 1) * 32].sum()) for i in range(min(T // 32, 8))])
            b_, t_, h_, d_ = idx[0].tolist()
            print("  sample o1:", o1[b_, t_, h_, d_:d_ + 4].tolist(), "o0:", o0[b_, t_, h_, d_:d_ + 4].tolist())
