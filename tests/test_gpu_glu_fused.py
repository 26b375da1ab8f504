"""-m gpu: the gated activation fused with the LoRA skinny products (uamd_glu_fwd_xa / uamd_glu_bwd_xa, csrc/glu.hip):
element-wise outputs BIT-IDENTICAL to the plain activation kernels, the rank products equal to the separate
uamd_lora_xa2 launches up to fp32 summation order, and the whole LoRA_MLP block unchanged within that."""
import pytest
import torch

from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _proj(n_out, n_in, r, g, dtype):
    W = (torch.randn(n_out, n_in, generator=g) * 0.02).to(dtype).to(DEV)
    A = (torch.randn(r, n_in, generator=g) * 0.02).to(DEV)
    B = (torch.randn(n_out, r, generator=g) * 0.02).to(DEV)
    return (W, None, A, B, 2.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["swiglu", "geglu_exact", "geglu_approx"])
@pytest.mark.parametrize("M,K,r", [(2048, 14336, 16), (300, 1024, 8), (17, 264, 32), (4096, 5632, 64)])
def test_fused_activation_and_rank_products(M, K, r, act, dtype):
    from unsloth_amd.kernels import geglu, swiglu, utils as U
    monkey = U.GLU_FUSED
    U.GLU_FUSED = "all"
    try:
        _fused_case(M, K, r, act, dtype, geglu, swiglu, U)
    finally:
        U.GLU_FUSED = monkey


def _fused_case(M, K, r, act, dtype, geglu, swiglu, U):
    fwd = {"swiglu": swiglu.swiglu_fg_kernel, "geglu_exact": geglu.geglu_exact_forward_kernel,
           "geglu_approx": geglu.geglu_approx_forward_kernel}[act]
    bwd = {"swiglu": swiglu.swiglu_DWf_DW_dfg_kernel, "geglu_exact": geglu.geglu_exact_backward_kernel,
           "geglu_approx": geglu.geglu_approx_backward_kernel}[act]
    g_ = torch.Generator().manual_seed(M + K)
    e = torch.randn(M, K, generator=g_).to(dtype).to(DEV)
    g = torch.randn(M, K, generator=g_).to(dtype).to(DEV)
    DW = (torch.randn(M, K, generator=g_) * 0.1).to(dtype).to(DEV)
    H = 512
    down = _proj(H, K, r, g_, dtype)
    up, gate = _proj(K, H, r, g_, dtype), _proj(K, H, r, g_, dtype)
    # forward
    out = U.glu_fwd_xa(act, e, g, down)
    if act != "swiglu":                       # only the hot path's activation takes the fused kernels
        assert out is None and U.glu_bwd_terms(act, DW.clone(), e.clone(), g.clone(), up, gate) is None
        return
    assert out is not None
    h, (xa, offs, xk) = out
    h_ref = fwd(e.clone(), g.clone())
    assert torch.equal(h, h_ref)
    xa_ref, offs_ref, xk_ref = U._xa_and_rank_block(h_ref, [down[2]], xk is not None)
    assert offs == offs_ref
    truth = h_ref.float() @ down[2].to(dtype).float().t()
    assert rel_fro(xa, truth) <= 2 * rel_fro(xa_ref, truth) + 1e-6
    assert rel_fro(xa, xa_ref) < 1e-5
    if xk is not None:
        assert xk.shape == xk_ref.shape and torch.all(xk[:, r:] == 0)
        assert rel_fro(xk.float(), xk_ref.float()) < 2e-3
    # backward
    DW1, e1, g1 = DW.clone(), e.clone(), g.clone()
    res = U.glu_bwd_terms(act, DW1, e1, g1, up, gate)
    assert res is not None
    h2, df, de, (pu, pg) = res
    DW2, e2, g2 = bwd(DW.clone(), e.clone(), g.clone())
    assert torch.equal(h2, DW2) and torch.equal(df, e2) and torch.equal(de, g2)
    tu, tg = U.lora_dx_terms([e2, g2], [up, gate])
    assert rel_fro(pu, tu) < 1e-5 and rel_fro(pg, tg) < 1e-5
    ku, kg = getattr(pu, "_uamd_xk", None), getattr(pg, "_uamd_xk", None)
    ru_, rg_ = getattr(tu, "_uamd_xk", None), getattr(tg, "_uamd_xk", None)
    assert (ku is None) == (ru_ is None)
    if ku is not None:
        assert ku[0] is kg[0] and ku[1] == ru_[1] and kg[1] == rg_[1] and ku[0].shape == ru_[0].shape
        assert torch.all(ku[0][:, 2 * r:] == 0)
        assert rel_fro(ku[0].float(), ru_[0].float()) < 2e-3


def test_lora_mlp_block_with_and_without_the_fusion():
    from unsloth_amd.kernels import utils as U
    from unsloth_amd.kernels.fast_lora import LoRA_MLP
    from unsloth_amd.kernels.swiglu import swiglu_DWf_DW_dfg_kernel, swiglu_fg_kernel
    g_ = torch.Generator().manual_seed(0)
    T, H, I, r = 2048, 1024, 2816, 16
    dtype = torch.bfloat16
    gate, up, down = _proj(I, H, r, g_, dtype), _proj(I, H, r, g_, dtype), _proj(H, I, r, g_, dtype)
    X = torch.randn(1, T, H, generator=g_).to(dtype).to(DEV)
    dY = (torch.randn(1, T, H, generator=g_) * 0.1).to(dtype).to(DEV)
    res = {}
    keep = U.GLU_FUSED
    for fused in ("all", False):
        U.GLU_FUSED = fused
        try:
            ps = [torch.nn.Parameter(t.clone()) for p in (gate, up, down) for t in (p[2], p[3])]
            x = X.clone().requires_grad_(True)
            out = LoRA_MLP.apply(x, gate[0], None, ps[0], ps[1], 2.0, up[0], None, ps[2], ps[3], 2.0, down[0], None, ps[4],
                                 ps[5], 2.0, swiglu_fg_kernel, swiglu_DWf_DW_dfg_kernel, False)
            out.backward(dY.clone())
            res[fused] = [out.detach().float(), x.grad.float()] + [p.grad.float() for p in ps]
        finally:
            U.GLU_FUSED = keep
    for a, b in zip(res["all"], res[False]):
        assert rel_fro(a, b) < 3e-3


@pytest.mark.parametrize("M,K,r", [(2048, 14336, 16), (300, 1024, 8), (17, 264, 32), (33, 5632, 64), (5, 8, 4), (64, 520, 16)])
def test_fused_activation_schedules_agree(M, K, r):
    """UAMD_TUNE_GLU_XA (knob 10): 0 = 4 waves per 16-row block (rounds 3-4), 1 = 8 waves, 2 = 8 waves + tiles requested two steps
    ahead, 3 = 2 + the columns of a row group split over adjacent workgroups (partial rank products summed in part order by the
    last workgroup of the row group) where the shape rule says so: default; 8 = the split always. The element-wise outputs are the same arithmetic on the same operands: BIT-IDENTICAL; the rank products sum
    the same tile products over 4 resp. 8 partial accumulators: equal up to fp32 summation order. K = 264 / 520 / 8: one, three
    and a fraction of a 256-column tile (every remainder branch of the depth-2 loop: 1, 2, 3, 4 and 56 = 3 * 17 + 5 tiles)."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels import utils as U
    L = _lib.lib()
    dtype = torch.bfloat16
    g_ = torch.Generator().manual_seed(M * 7 + K)
    e = torch.randn(M, K, generator=g_).to(dtype).to(DEV)
    g = torch.randn(M, K, generator=g_).to(dtype).to(DEV)
    DW = (torch.randn(M, K, generator=g_) * 0.1).to(dtype).to(DEV)
    H = 512
    down, up, gate = _proj(H, K, r, g_, dtype), _proj(K, H, r, g_, dtype), _proj(K, H, r, g_, dtype)
    keep = U.GLU_FUSED
    U.GLU_FUSED = "all"
    res = {}
    try:
        for v in (0, 1, 2, 3, 8):
            assert L.uamd_set_tuning(10, v) == 0
            out = U.glu_fwd_xa("swiglu", e, g, down)
            if out is None:
                pytest.skip("shape not fusable")
            h, (xa, _, xk) = out
            h2, df, de, (pu, pg) = U.glu_bwd_terms("swiglu", DW.clone(), e.clone(), g.clone(), up, gate)
            res[v] = (h, h2, df, de, xa, pu, pg)
    finally:
        L.uamd_set_tuning(10, 3)
        U.GLU_FUSED = keep
    for v in (1, 2, 3, 8):
        for a, b in zip(res[0][:4], res[v][:4]):
            assert torch.equal(a, b), v
        for a, b in zip(res[0][4:], res[v][4:]):
            assert rel_fro(a, b) < 1e-5, v
    for a, b in zip(res[1][4:], res[2][4:]):                  # the same 8 partial sums, the same order: bitwise
        assert torch.equal(a, b)


def test_column_split_is_run_to_run_deterministic_and_leaves_its_counters_zero():
    """The workgroup that finishes a row group LAST differs from run to run; the sum it forms does not (tile order). And the
    arrival counters are zero again after every launch -- the next launch depends on it."""
    from unsloth_amd.kernels import utils as U
    dtype = torch.bfloat16
    g_ = torch.Generator().manual_seed(5)
    M, K, r = 1000, 14336, 16
    e = torch.randn(M, K, generator=g_).to(dtype).to(DEV)
    g = torch.randn(M, K, generator=g_).to(dtype).to(DEV)
    DW = (torch.randn(M, K, generator=g_) * 0.1).to(dtype).to(DEV)
    down, up, gate = _proj(512, K, r, g_, dtype), _proj(K, 512, r, g_, dtype), _proj(K, 512, r, g_, dtype)
    keep = U.GLU_FUSED
    U.GLU_FUSED = "all"
    try:
        runs = []
        for _ in range(4):
            h, (xa, _, xk) = U.glu_fwd_xa("swiglu", e, g, down)
            _, _, _, (pu, pg) = U.glu_bwd_terms("swiglu", DW.clone(), e.clone(), g.clone(), up, gate)
            runs.append((xa.clone(), pu.clone(), pg.clone()))
        torch.cuda.synchronize()
        for other in runs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(runs[0], other))
        assert U._GLU_WS and all(int(c.abs().sum()) == 0 for _, c in U._GLU_WS.values())
    finally:
        U.GLU_FUSED = keep
