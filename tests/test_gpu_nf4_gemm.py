"""-m gpu: NF4 quantise/dequantise (bit-exact vs the oracle's restatement of the bitsandbytes
format), the MFMA GEMMs (dense, fused-NF4, grouped, LoRA-fused, accumulate) and the skinny XA
kernel against fp32 references of the same bf16/fp16 inputs."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref_ops as R
from tests._util import assert_ulp, rel_fro

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def test_mfma_probe_layout():
    """the (lane,reg)->(row,col) map the GEMM epilogue relies on (cdna guide section 3)."""
    from unsloth_amd import _lib
    out = torch.zeros(2, 64, 4, device=DEV)
    _lib.check(_lib.lib().uamd_debug_mfma_probe(_lib.ptr(out), _lib.stream_of(out)), "probe")
    out = out.cpu()
    lanes = torch.arange(64)
    want_row = ((lanes >> 4) * 4)[:, None] + torch.arange(4)[None, :] + 1     # row = 4*(lane>>4)+reg
    want_col = (lanes & 15)[:, None].expand(64, 4) + 1                          # col = lane&15
    assert torch.equal(out[0].round().long(), want_row), out[0]
    assert torch.equal(out[1].round().long(), want_col), out[1]


# ---------------------------------------------------------------- NF4
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_nf4_quantize_matches_oracle_bit_exact(dtype):
    from unsloth_amd.nf4 import quantize_nf4
    W = (torch.randn(96, 256, generator=g(1)) * 0.02).to(dtype)
    W[5, :64] = 0                                   # an all-zero block
    packed, qs = quantize_nf4(W.to(DEV), compress_statistics=False)
    p_ref, a_ref = R.nf4_quantize_np(W.float().numpy().reshape(-1), 64)
    assert np.array_equal(packed.cpu().numpy().reshape(-1), p_ref)
    assert np.array_equal(qs.absmax.cpu().numpy(), a_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("nested", [False, True])
def test_nf4_dequantize_bit_exact(dtype, nested):
    from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4
    from unsloth_amd.kernels import fast_dequantize
    W = (torch.randn(192, 320, generator=g(2)) * 0.02).to(dtype)
    packed, qs = quantize_nf4(W.to(DEV), compress_statistics=nested)
    want = R.nf4_dequantize_state(packed, qs)                     # CPU restatement, same bytes
    for cache in (False, True):
        got = dequantize_nf4(packed, qs, cache_absmax=cache)
        assert torch.equal(got.cpu(), want), f"dequant mismatch (cache_absmax={cache})"
    if dtype != torch.float32:
        got_t = dequantize_nf4(packed, qs, transpose=True)
        assert torch.equal(got_t.cpu(), want.t())
    # reference call conventions: passthrough, and W.t() -> transposed result (utils.py:578, 678)
    assert fast_dequantize(W, None) is W
    assert torch.equal(fast_dequantize(packed.t(), qs).cpu(), want.t())
    # quantisation error sanity: NF4 with absmax scaling keeps |err| <= absmax * max half-gap
    err = (want.float() - W.float()).abs().max()
    assert err < 0.16 * W.float().abs().max()


@pytest.mark.parametrize("blocksize", [32, 64, 256])
@pytest.mark.parametrize("nested", [False, True])
@pytest.mark.parametrize("rows,cols", [(128, 64), (192, 320), (520, 1032), (4096, 14336)])
def test_nf4_dequantize_four_groups_per_lane_kernel_is_bit_identical(rows, cols, nested, blocksize):
    """nf4_dequant_x4_kernel (knob 8, the default for 16-bit outputs of >= 8192 elements) against the one-group kernel
    and, at the sizes the CPU restatement finishes quickly, the oracle: whole trips only (128 x 64 = one trip), trips + a
    group remainder, absmax blocks that straddle trips (blocksize 256), nested and plain statistics, the [14336, 4096]
    weight of the step (7168 trips over 1792 blocks)."""
    from unsloth_amd import _lib
    from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4
    if (rows * cols) % blocksize:
        pytest.skip("numel not a multiple of the blocksize")
    for dtype in (torch.bfloat16, torch.float16):
        W = (torch.randn(rows, cols, generator=g(rows + cols)) * 0.02).to(dtype)
        packed, qs = quantize_nf4(W.to(DEV), blocksize=blocksize, compress_statistics=nested)
        L = _lib.lib()
        try:
            assert L.uamd_set_tuning(8, 0) == 0
            one = dequantize_nf4(packed, qs, cache_absmax=False).clone()
            assert L.uamd_set_tuning(8, 1) == 0
            four = dequantize_nf4(packed, qs, cache_absmax=False).clone()
            four_cached = dequantize_nf4(packed, qs, cache_absmax=True).clone()
        finally:
            L.uamd_set_tuning(8, 1)
        assert torch.equal(one, four) and torch.equal(one, four_cached)
        if rows * cols <= 600000:
            assert torch.equal(four.cpu(), R.nf4_dequantize_state(packed, qs))


@pytest.mark.parametrize("nested", [False, True])
@pytest.mark.parametrize("shapes", [[(4096, 4096), (1024, 4096), (1024, 4096)], [(14336, 4096), (14336, 4096)],
                                    [(64, 128), (192, 128), (64, 128), (128, 128), (64, 128)], [(64, 128), (96, 264)]])
def test_nf4_dequantize_group_launch_is_bit_identical(shapes, nested):
    """uamd_nf4_dequantize_multi (nf4.dequantize_nf4_group): the weights of one grouped GEMM decoded by one launch per four
    -- q | k | v and gate | up at the step's sizes, five small weights (two launches), trips that straddle segment boundaries
    inside a block -- against the single launches; a member the grouped launch does not take (numel not a multiple of 8192)
    sends the whole group through the single launches."""
    from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4, dequantize_nf4_group
    for dtype in (torch.bfloat16, torch.float16):
        pks, qss, want = [], [], []
        for i, (r, c) in enumerate(shapes):
            W = (torch.randn(r, c, generator=g(r + c + i)) * 0.02).to(dtype)
            pk, qs = quantize_nf4(W.to(DEV), compress_statistics=nested)
            pks.append(pk)
            qss.append(qs)
            want.append(dequantize_nf4(pk, qs).clone())
        # members of equal width as row blocks of one buffer (how the stacked decode of the dX product uses it), others alone
        same = len({c for _, c in shapes}) == 1
        buf = torch.full((sum(r for r, _ in shapes), shapes[0][1]), 7.0, dtype=dtype, device=DEV) if same else None
        outs, row = [], 0
        for (r, c) in shapes:
            outs.append(buf[row:row + r] if same else torch.empty((r, c), dtype=dtype, device=DEV))
            row += r
        got = dequantize_nf4_group(pks, qss, outs)
        for a, b in zip(got, want):
            assert torch.equal(a, b)


@pytest.mark.parametrize("rows,cols", [(1024, 4096), (200, 264), (64, 256), (136, 1032)])
@pytest.mark.parametrize("knob", [0, 1])
def test_nf4_dequantize_transposed_kernels_bit_exact(rows, cols, knob):
    """Both transposing kernels (64x64 tile, 64x256 tile) against the CPU restatement; ragged rows/cols, absmax
    blocks that straddle W rows (cols % 64 != 0), nested statistics."""
    from unsloth_amd import _lib
    from unsloth_amd.nf4 import quantize_nf4, dequantize_nf4
    W = (torch.randn(rows, cols, generator=g(4)) * 0.02).to(torch.bfloat16)
    packed, qs = quantize_nf4(W.to(DEV), compress_statistics=True)
    want = R.nf4_dequantize_state(packed, qs)
    L = _lib.lib()
    try:
        assert L.uamd_set_tuning(3, knob) == 0
        got_t = dequantize_nf4(packed, qs, transpose=True)
        assert torch.equal(got_t.cpu(), want.t())
    finally:
        L.uamd_set_tuning(3, 1)


def test_bnb_compat_entry_points():
    """The symbols the reference binds (unsloth/kernels/utils.py:272-275), called the way
    fast_dequantize calls them (:650-675): nested absmax -> += offset -> NF4."""
    from unsloth_amd import _lib
    from unsloth_amd.nf4 import quantize_nf4
    W = (torch.randn(64, 512, generator=g(3)) * 0.02).to(torch.bfloat16)
    packed, qs = quantize_nf4(W.to(DEV), compress_statistics=True)
    n_abs = qs.absmax.numel()
    out_absmax = torch.empty(n_abs, dtype=torch.float32, device=DEV)
    L = _lib.lib()
    st = _lib.stream_of(out_absmax)
    L.cdequantize_blockwise_fp32(_lib.ptr(qs.state2.code), _lib.ptr(qs.absmax), _lib.ptr(qs.state2.absmax),
                                 _lib.ptr(out_absmax), ctypes.c_int(qs.state2.blocksize), ctypes.c_int(n_abs), st)
    out_absmax += qs.offset
    out = torch.empty(64, 512, dtype=torch.bfloat16, device=DEV)
    L.cdequantize_blockwise_bf16_nf4(None, _lib.ptr(packed), _lib.ptr(out_absmax), _lib.ptr(out),
                                     ctypes.c_int(qs.blocksize), ctypes.c_int(out.numel()), st)
    want = R.nf4_dequantize_state(packed, qs)
    # `+= offset` on the GPU is one fp32 add, the oracle does code*absmax2 + offset the same way
    assert torch.equal(out.cpu(), want)


# ---------------------------------------------------------------- GEMM
def _ref_mm(X, W):
    return X.float() @ W.float().t()


def _check_gemm(got, want_f32, dtype, K, what):
    """fp32-accumulated result rounded once: within 1 ulp of the fp32 reference + accumulation noise."""
    eps = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    scale = float(want_f32.abs().mean())
    assert_ulp(got, want_f32, dtype, ulps=1, atol=eps * scale * 0.5 + 1e-6 * scale * K ** 0.5, what=what,
               allow_frac=1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 130, 136), (1, 128, 4096),
                                   (2048, 1024, 4096), (333, 4096, 1088)])
def test_gemm_nt_dense(dtype, M, N, K):
    from unsloth_amd.kernels.utils import lora_linear_forward
    X = torch.randn(M, K, generator=g(5)).to(dtype)
    W = (torch.randn(N, K, generator=g(6)) * 0.05).to(dtype)
    (Y,) = lora_linear_forward(X.to(DEV), [(W.to(DEV), None, None, None, None)])
    _check_gemm(Y, _ref_mm(X, W), dtype, K, f"gemm {M}x{N}x{K}")


def test_gemm_transpose_detecting():
    """asymmetric operands: catches an output written as C^T or a swapped fragment map."""
    from unsloth_amd.kernels.utils import lora_linear_forward
    M, N, K = 128, 256, 64
    X = torch.zeros(M, K)
    W = torch.zeros(N, K)
    X[:, 0] = torch.arange(M).float() % 7 + 1
    W[:, 0] = torch.arange(N).float() % 5 + 1
    X[:, 1] = 1
    W[:, 1] = (torch.arange(N) % 3).float()
    (Y,) = lora_linear_forward(X.to(torch.bfloat16).to(DEV), [(W.to(torch.bfloat16).to(DEV), None, None, None, None)])
    assert torch.equal(Y.float().cpu(), X @ W.t())


@pytest.mark.parametrize("r", [8, 16, 64])
def test_gemm_with_lora_and_groups(r):
    from unsloth_amd.kernels.utils import lora_linear_forward
    dtype = torch.bfloat16
    M, K = 300, 512
    Ns = [512, 128, 128]
    X = torch.randn(M, K, generator=g(7)).to(dtype)
    projs, refs = [], []
    for i, N in enumerate(Ns):
        W = (torch.randn(N, K, generator=g(10 + i)) * 0.05).to(dtype)
        A = torch.randn(r, K, generator=g(20 + i)) * 0.05
        B = torch.randn(N, r, generator=g(30 + i)) * 0.05
        s = 2.0
        projs.append((W.to(DEV), None, A.to(DEV), B.to(DEV), s))
        xa = (X.float() @ A.to(dtype).float().t()).to(dtype).float()
        refs.append(_ref_mm(X, W) + s * xa @ B.to(dtype).float().t())
    outs = lora_linear_forward(X.to(DEV), projs)
    for o, ref, N in zip(outs, refs, Ns):
        _check_gemm(o, ref, dtype, K, f"grouped lora N={N} r={r}")


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (2048, 4096, 4096), (77, 1024, 256), (512, 14336, 4096)])
def test_gemm_nf4_fused_matches_dequant_then_gemm(M, N, K):
    from unsloth_amd.nf4 import quantize_nf4
    from unsloth_amd.kernels import utils as U
    dtype = torch.bfloat16
    X = torch.randn(M, K, generator=g(40)).to(dtype).to(DEV)
    W = (torch.randn(N, K, generator=g(41)) * 0.02).to(dtype).to(DEV)
    packed, qs = quantize_nf4(W, compress_statistics=True)
    Wd = R.nf4_dequantize_state(packed, qs)                       # oracle-decoded weight
    want = _ref_mm(X.cpu(), Wd)
    old = (U.FUSED_NF4, U.FUSED_NF4_MAX_M)
    try:
        for fused in (True, False):
            U.FUSED_NF4, U.FUSED_NF4_MAX_M = fused, 1 << 30        # force the decode-in-GEMM kernel at any M
            (Y,) = U.lora_linear_forward(X, [(packed, qs, None, None, None)])
            _check_gemm(Y, want, dtype, K, f"nf4 gemm fused={fused} {M}x{N}x{K}")
        U.FUSED_NF4, U.FUSED_NF4_MAX_M = old                       # the shipped policy picks by M
        (Y,) = U.lora_linear_forward(X, [(packed, qs, None, None, None)])
        _check_gemm(Y, want, dtype, K, f"nf4 gemm policy {M}x{N}x{K}")
    finally:
        U.FUSED_NF4, U.FUSED_NF4_MAX_M = old


def test_lora_linear_dx_accumulates_groups():
    from unsloth_amd.nf4 import quantize_nf4
    from unsloth_amd.kernels.utils import lora_linear_dx
    dtype = torch.bfloat16
    M, Kin, r = 260, 512, 16
    Ns = [512, 128]
    want = torch.zeros(M, Kin)
    dYs, projs = [], []
    for i, N in enumerate(Ns):
        W = (torch.randn(N, Kin, generator=g(50 + i)) * 0.02).to(dtype).to(DEV)
        packed, qs = quantize_nf4(W, compress_statistics=True)
        Wd = R.nf4_dequantize_state(packed, qs).float()
        A = torch.randn(r, Kin, generator=g(60 + i)) * 0.05
        B = torch.randn(N, r, generator=g(70 + i)) * 0.05
        dY = torch.randn(M, N, generator=g(80 + i)).to(dtype)
        s = 0.5
        dyb = (dY.float() @ B.to(dtype).float()).to(dtype).float()
        want = want + dY.float() @ Wd + s * dyb @ A.to(dtype).float()
        dYs.append(dY.to(DEV))
        projs.append((packed, qs, A.to(DEV), B.to(DEV), s))
    out = torch.full((M, Kin), 7.0, dtype=dtype, device=DEV)      # must be overwritten, not added to
    got = lora_linear_dx(dYs, projs, out=out)
    assert got.data_ptr() == out.data_ptr()
    # two sequential roundings (per group) -> 2 ulp
    assert_ulp(got, want, dtype, ulps=2, atol=2.0 ** -7 * float(want.abs().mean()), what="dX", allow_frac=1e-3)


@pytest.mark.parametrize("M,K,Rs", [(2048, 4096, [16, 16, 16]), (100, 512, [8]), (64, 14336, [64, 64]), (333, 520, [16, 8]),
                                    (8192, 1024, [16]), (33, 64, [64]), (1, 8, [4])])
def test_lora_xa(M, K, Rs):
    from unsloth_amd.kernels.utils import lora_xa
    dtype = torch.bfloat16
    X = torch.randn(M, K, generator=g(90)).to(dtype)
    As = [torch.randn(r, K, generator=g(91 + i)) * 0.05 for i, r in enumerate(Rs)]
    out, offs = lora_xa(X.to(DEV), [a.to(DEV) for a in As])
    for a, (o, rp) in zip(As, offs):
        want = X.float() @ a.to(dtype).float().t()
        got = out[:, o:o + a.shape[0]].cpu()
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4 * float(want.abs().mean()) + 1e-5)
        assert torch.all(out[:, o + a.shape[0]:o + rp] == 0)
    # deterministic split-K: two runs are bitwise identical
    out2, _ = lora_xa(X.to(DEV), [a.to(DEV) for a in As])
    assert torch.equal(out, out2)


# ---------------------------------------------------------------- 256x256 LDS-DMA ping-pong kernel
@pytest.fixture
def force256():
    """Force the 256x256 kernel (csrc/gemm256.hip) for every shape."""
    from unsloth_amd.kernels import utils as U
    old = U.GEMM256_MODE
    U.GEMM256_MODE = "on"
    yield "pp"
    U.GEMM256_MODE = old


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 768, 256), (300, 260, 192), (1, 256, 4096),
                                   (4096, 4096, 4096), (1000, 1024, 14336), (260, 516, 32), (256, 256, 96),
                                   (777, 333, 160)])
def test_gemm256_dense(force256, dtype, M, N, K):
    if K % 64:
        pytest.skip("gemm256.hip needs K % 64 == 0 (the dispatcher routes other K to the 128x128 kernel)")
    from unsloth_amd.kernels.utils import lora_linear_forward
    X = torch.randn(M, K, generator=g(105)).to(dtype)
    W = (torch.randn(N, K, generator=g(106)) * 0.05).to(dtype)
    (Y,) = lora_linear_forward(X.to(DEV), [(W.to(DEV), None, None, None, None)])
    _check_gemm(Y, _ref_mm(X, W), dtype, K, f"gemm256 {M}x{N}x{K}")
    # run-to-run bitwise determinism (catches LDS races that only sometimes bite)
    for _ in range(3):
        (Y2,) = lora_linear_forward(X.to(DEV), [(W.to(DEV), None, None, None, None)])
        assert torch.equal(Y, Y2)


@pytest.mark.parametrize("half", [0, 2])
@pytest.mark.parametrize("M,N,K", [(128, 256, 64), (128, 256, 128), (2048, 4096, 4096), (300, 516, 192), (2048, 1024, 14336),
                                   (1, 256, 256), (129, 255, 320)])
def test_gemm256_tile_heights(force256, half, M, N, K):
    """128 x 256 tiles (three-stage ring, csrc/gemm256.hip gemm_nt256h_kernel) and 256 x 256 tiles give the same
    fp32-accumulated product; rank block included; run-to-run bitwise deterministic."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels.utils import lora_linear_forward
    dtype = torch.bfloat16
    X = torch.randn(M, K, generator=g(151)).to(dtype)
    W = (torch.randn(N, K, generator=g(152)) * 0.05).to(dtype)
    A = torch.randn(16, K, generator=g(153)) * 0.05
    B = torch.randn(N, 16, generator=g(154)) * 0.05
    L = _lib.lib()
    try:
        L.uamd_set_tuning(6, half)
        (Y,) = lora_linear_forward(X.to(DEV), [(W.to(DEV), None, None, None, None)])
        _check_gemm(Y, _ref_mm(X, W), dtype, K, f"gemm256 half={half} {M}x{N}x{K}")
        (Y2,) = lora_linear_forward(X.to(DEV), [(W.to(DEV), None, None, None, None)])
        assert torch.equal(Y, Y2)
        (Z,) = lora_linear_forward(X.to(DEV), [(W.to(DEV), None, A.to(DEV), B.to(DEV), 2.0)])
        xa = (X.float() @ A.to(dtype).float().t()).to(dtype).float()
        _check_gemm(Z, _ref_mm(X, W) + xa @ (2.0 * B).to(dtype).float().t(), dtype, K, f"gemm256+lora half={half}")
    finally:
        L.uamd_set_tuning(6, 1)


@pytest.mark.parametrize("half", [0, 2])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 520, 192), (2048, 4096, 1024), (1000, 1024, 6144), (128, 8, 64),
                                   (77, 4104, 128)])
def test_gemm256_nn_form(half, dtype, M, N, K):
    """uamd_gemm_nn_256: C = A @ B with B [K, N] row-major (the dX products: contraction over the weight's ROWS through
    transposing LDS reads), both tile heights, rank block as [Rk, N] extra K tiles, accumulate, ragged M / N,
    bitwise run-to-run determinism; and it equals the NT form fed the transposed operand bit for bit."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels.utils import _group, _launch_gemm, rank_block_bk
    X = torch.randn(M, K, generator=g(161)).to(dtype)
    B = (torch.randn(K, N, generator=g(162)) * 0.05).to(dtype)
    P = torch.randn(M, 16, generator=g(163))
    Al = torch.randn(16, N, generator=g(164)) * 0.05
    L = _lib.lib()
    Xd, Bd = X.to(DEV), B.to(DEV)
    try:
        L.uamd_set_tuning(6, half)
        C = torch.empty(M, N, dtype=dtype, device=DEV)
        _launch_gemm(Xd, [_group(Bd, C, N, Bd.stride(0))], nf4=False, nn=True)
        ref = X.float() @ B.float()
        _check_gemm(C, ref, dtype, K, f"gemm_nn {M}x{N}x{K} half={half}")
        C2 = torch.empty_like(C)
        _launch_gemm(Xd, [_group(Bd, C2, N, Bd.stride(0))], nf4=False, nn=True)
        assert torch.equal(C, C2)
        # same product through the NT form on B^T: identical accumulation order -> identical bits
        Bt = Bd.t().contiguous()
        C3 = torch.empty_like(C)
        from unsloth_amd.kernels import utils as U
        old = U.GEMM256_MODE
        U.GEMM256_MODE = "on"
        try:
            _launch_gemm(Xd, [_group(Bt, C3, N, Bt.stride(0))], nf4=False)
        finally:
            U.GEMM256_MODE = old
        assert torch.equal(C, C3)
        # rank block: XK = T(P) zero-padded to 64 columns, BK = [2.0 * Al; 0] as [64, N]; then accumulate on top
        xk = torch.zeros(M, 64, dtype=dtype, device=DEV)
        xk[:, :16] = P.to(DEV)
        bk = rank_block_bk([(Al.to(DEV), 0, 2.0)], 64, N, False, dtype, by_rows=True)
        C4 = torch.empty_like(C)
        _launch_gemm(Xd, [_group(Bd, C4, N, Bd.stride(0), xa=P.to(DEV), ld_xa=16, R=16, scale=2.0, xk=xk, bk=bk)],
                     nf4=False, nn=True)
        ref4 = ref + P.to(dtype).float() @ (2.0 * Al).to(dtype).float()
        _check_gemm(C4, ref4, dtype, K, "gemm_nn + rank block")
        C5 = C.clone()
        _launch_gemm(Xd, [_group(Bd, C5, N, Bd.stride(0))], nf4=False, accumulate=True, nn=True)
        _check_gemm(C5, C.float().cpu() + ref, dtype, K, "gemm_nn accumulate")
    finally:
        L.uamd_set_tuning(6, 1)


@pytest.mark.parametrize("nn", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,Ns,K,rank", [(4096, (4096, 1024, 1024), 512, True),      # 3 groups, odd K-tile count (8 + 1)
                                         (4096 + 40, (4096 + 8,), 256, False),        # ragged M and N, even count (4)
                                         (8192, (2560,), 1024, True),                  # 320 tiles: CUs get 1 or 2
                                         (2048, (8192 + 256,), 448, True)])            # 7 + 1 K tiles
def test_gemm256_persistent_walk_is_bit_identical(nn, dtype, M, Ns, K, rank):
    """gemm_nt256p_kernel (one block per CU walks the tiles, prefetching the next output tile's first K tiles during the
    current one's last and swapping LDS stage addresses after an odd K-tile count) against gemm_nt256_kernel (one block
    per tile): same per-tile accumulation order, so bit-identical -- multi-group launches with per-group rank blocks,
    ragged edges, accumulate, both operand forms; run-to-run deterministic."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels.utils import _group, _launch_gemm
    from unsloth_amd.kernels import utils as U
    L = _lib.lib()
    X = torch.randn(M, K, generator=g(171)).to(dtype).to(DEV)
    xk = torch.zeros(M, 64, dtype=dtype, device=DEV)
    xk[:, :16] = torch.randn(M, 16, generator=g(172)).to(dtype).to(DEV)
    Bs, BKs = [], []
    for i, N in enumerate(Ns):
        Bs.append(((torch.randn(K, N, generator=g(173 + i)) if nn else torch.randn(N, K, generator=g(173 + i))) * 0.05).to(dtype).to(DEV))
        bk = torch.zeros((64, N) if nn else (N, 64), dtype=dtype, device=DEV)
        blk = (torch.randn(16, N, generator=g(183 + i)) * 0.05).to(dtype).to(DEV)
        if nn:
            bk[:16] = blk
        else:
            bk[:, :16] = blk.t()
        BKs.append(bk)

    def run(persist, accumulate):
        outs = [torch.full((M, N), 0.25, dtype=dtype, device=DEV) for N in Ns]
        groups = []
        for i, N in enumerate(Ns):
            use_rank = rank and i != 1          # the middle group of a 3-group launch goes without: per-group K-tile counts
            groups.append(_group(Bs[i], outs[i], N, Bs[i].stride(0), xa=xk if use_rank else None, ld_xa=64, R=16, scale=1.0,
                                 xk=xk if use_rank else None, bk=BKs[i] if use_rank else None))
        L.uamd_set_tuning(7, 2 if persist else 0)
        _launch_gemm(X, groups, nf4=False, accumulate=accumulate, nn=nn)
        return outs

    old = U.GEMM256_MODE
    U.GEMM256_MODE = "on"
    try:
        L.uamd_set_tuning(6, 0)                  # 256-row tiles (the persistent walk exists for those)
        L.uamd_set_tuning(11, 0)                 # the 8-wave kernels (whole-tile NT launches would take gemm_nt256s_kernel)
        for accumulate in (False, True):
            ref = run(0, accumulate)
            for plain in (1, 0):                 # knob 9: the load-free-epilogue instance (default) / the run-time-dispatch one
                L.uamd_set_tuning(9, plain)
                got = run(1, accumulate)
                for r, o in zip(ref, got):
                    assert torch.equal(r, o)
            L.uamd_set_tuning(9, 1)
        # and against the fp32 product
        y = run(1, False)[0]
        Xf = X.float().cpu()
        want = Xf @ (Bs[0].float().cpu() if nn else Bs[0].float().cpu().t())
        if rank:
            want = want + xk.float().cpu() @ (BKs[0].float().cpu() if nn else BKs[0].float().cpu().t())
        _check_gemm(y, want, dtype, K, "persistent gemm256")
    finally:
        U.GEMM256_MODE = old
        L.uamd_set_tuning(6, 1)
        L.uamd_set_tuning(7, 1)
        L.uamd_set_tuning(9, 1)
        L.uamd_set_tuning(11, 1)


@pytest.mark.parametrize("nn", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,Ns,K,rank", [(4096, (4096, 1024, 1024), 512, True),      # 3 groups, per-group rank blocks
                                         (512, (256,), 128, False),                    # below the kernel's K >= 192: the 8-wave path
                                         (512, (512,), 192, True),                     # one loop trip + one rank tile
                                         (2048, (2560,), 1024, True),
                                         (1024, (1024,), 4096, False),                 # a long K loop
                                         (256, (768, 256), 320, True),                 # odd K-tile count
                                         (4096, (8192,), 320, True),                   # 512 tiles: the persistent walk, 2 per CU, odd count
                                         (2048, (16384 + 256, 256), 192, True),        # 528 tiles over 256 CUs: uneven walk, 2 groups
                                         (8192, (4096,), 256, False)])                 # 512 tiles, even K-tile count (no stage flip)
def test_gemm256s_one_wave_per_simd_kernel_is_bit_identical(nn, dtype, M, Ns, K, rank):
    """gemm_nt256s_kernel (4 waves x 128 x 128, hand-ordered K loop, `buffer_load ... lds` pieces with SGPR offsets; knob 11,
    default for whole-tile NT and NN launches) against the 8-wave gemm_nt256_kernel: same LDS image, same per-accumulator k
    order (main tiles in order, then the rank block's), so BIT-IDENTICAL -- multi-group launches with per-group rank blocks,
    accumulate, bias-free, padded row strides, both operand forms (NN: B and BK as [K, N], transposing fragment reads into
    pinned registers); knob 11 = 9 takes the PERSISTENT walk from two output tiles per CU on (the next tile's first K tiles
    fetched during the current one's last, stage parity carried across tiles; the default, 1, from four on), 2 = one workgroup
    per tile always; and run-to-run deterministic."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels.utils import _group, _launch_gemm
    from unsloth_amd.kernels import utils as U
    L = _lib.lib()
    Xs = torch.empty(M, K + 64, dtype=dtype, device=DEV)[:, :K]          # lda != K
    Xs.copy_(torch.randn(M, K, generator=g(271)).to(dtype))
    xk = torch.zeros(M, 64, dtype=dtype, device=DEV)
    xk[:, :16] = torch.randn(M, 16, generator=g(272)).to(dtype).to(DEV)
    Bs, BKs = [], []
    for i, N in enumerate(Ns):
        W = (torch.randn(N, K, generator=g(273 + i)) * 0.05).to(dtype)
        blk = (torch.randn(N, 16, generator=g(283 + i)) * 0.05).to(dtype)
        if nn:                                                    # [K, N] with a padded row stride
            Bn = torch.empty(K, N + 8, dtype=dtype, device=DEV)[:, :N]
            Bn.copy_(W.t())
            Bs.append(Bn)
            bk = torch.zeros((64, N), dtype=dtype, device=DEV)
            bk[:16] = blk.t().to(DEV)
        else:
            Bs.append(W.to(DEV))
            bk = torch.zeros((N, 64), dtype=dtype, device=DEV)
            bk[:, :16] = blk.to(DEV)
        BKs.append(bk)

    def run(knob, accumulate):
        outs = [torch.full((M, N), 0.25, dtype=dtype, device=DEV) for N in Ns]
        groups = []
        for i, N in enumerate(Ns):
            use_rank = rank and i != 1
            groups.append(_group(Bs[i], outs[i], N, Bs[i].stride(0), xa=xk if use_rank else None, ld_xa=64, R=16, scale=1.0,
                                 xk=xk if use_rank else None, bk=BKs[i] if use_rank else None))
        L.uamd_set_tuning(11, knob)
        _launch_gemm(Xs, groups, nf4=False, accumulate=accumulate, nn=nn)
        return outs

    old = U.GEMM256_MODE
    U.GEMM256_MODE = "on"
    try:
        L.uamd_set_tuning(6, 0)
        L.uamd_set_tuning(7, 0)
        for accumulate in (False, True):
            ref = run(0, accumulate)
            got = run(9, accumulate)                 # the persistent walk from two tiles per CU on
            again = run(9, accumulate)
            single = run(2, accumulate)              # one workgroup per tile
            for r, o, o2, o3 in zip(ref, got, again, single):
                assert torch.equal(o, o2), "run-to-run"
                assert torch.equal(r, o), float((r.float() - o.float()).abs().max())
                assert torch.equal(r, o3), float((r.float() - o3.float()).abs().max())
        y = run(1, False)[0]
        want = Xs.float().cpu() @ (Bs[0].float().cpu() if nn else Bs[0].float().cpu().t())
        if rank:
            want = want + xk.float().cpu() @ (BKs[0].float().cpu() if nn else BKs[0].float().cpu().t())
        _check_gemm(y, want, dtype, K, "gemm256s")
    finally:
        U.GEMM256_MODE = old
        L.uamd_set_tuning(6, 1)
        L.uamd_set_tuning(7, 1)
        L.uamd_set_tuning(11, 1)


def test_gemm256_transpose_detecting_and_k_order(force256):
    from unsloth_amd.kernels.utils import lora_linear_forward
    M, N, K = 512, 512, 256
    X = torch.zeros(M, K)
    W = torch.zeros(N, K)
    for k in range(0, K, 37):                    # a few isolated k positions with distinct weights
        X[:, k] = (torch.arange(M) % 7 + 1).float() * (1 + k % 3)
        W[:, k] = (torch.arange(N) % 5 + 1).float() * (1 + k % 2)
    (Y,) = lora_linear_forward(X.to(torch.bfloat16).to(DEV), [(W.to(torch.bfloat16).to(DEV), None, None, None, None)])
    assert torch.equal(Y.float().cpu(), (X @ W.t()).to(torch.bfloat16).float())


def test_gemm256_groups_lora_accumulate(force256):
    from unsloth_amd.kernels.utils import lora_linear_forward, _group, _launch_gemm
    dtype = torch.bfloat16
    M, K, r = 700, 512, 16
    Ns = [512, 300, 128]
    X = torch.randn(M, K, generator=g(107)).to(dtype)
    projs, refs = [], []
    for i, N in enumerate(Ns):
        W = (torch.randn(N, K, generator=g(110 + i)) * 0.05).to(dtype)
        A = torch.randn(r, K, generator=g(120 + i)) * 0.05
        B = torch.randn(N, r, generator=g(130 + i)) * 0.05
        projs.append((W.to(DEV), None, A.to(DEV), B.to(DEV), 0.5))
        xa = (X.float() @ A.to(dtype).float().t()).to(dtype).float()
        refs.append(_ref_mm(X, W) + 0.5 * xa @ B.to(dtype).float().t())
    outs = lora_linear_forward(X.to(DEV), projs)
    for o, ref, N in zip(outs, refs, Ns):
        _check_gemm(o, ref, dtype, K, f"gemm256 grouped lora N={N}")
    # the same with the factors as fp32 Parameters: the rank block's BK operand then comes from the persistent,
    # once-per-step uamd_lora_prepare buffers (zero-padded, scale folded in) instead of per-call torch ops
    pprojs = [(W, q, torch.nn.Parameter(A), torch.nn.Parameter(B), s) for (W, q, A, B, s) in projs]
    outs_p = lora_linear_forward(X.to(DEV), pprojs)
    for o, o2 in zip(outs, outs_p):
        assert torch.equal(o, o2)
    with torch.no_grad():
        pprojs[1][3].mul_(2.0)                     # in-place update of one B: the prepared copy must follow
    from unsloth_amd.kernels.utils import invalidate_cast_cache
    invalidate_cast_cache()
    outs_q = lora_linear_forward(X.to(DEV), pprojs)
    assert torch.equal(outs_q[0], outs[0]) and torch.equal(outs_q[2], outs[2])
    xa1 = (X.float() @ projs[1][2].cpu().to(dtype).float().t()).to(dtype).float()
    _check_gemm(outs_q[1], _ref_mm(X, projs[1][0].cpu()) + 0.5 * xa1 @ pprojs[1][3].detach().cpu().to(dtype).float().t(),
                dtype, K, "gemm256 grouped lora after in-place update")
    # accumulate: C += A @ B^T
    C0 = torch.randn(M, Ns[0], generator=g(140)).to(dtype)
    C = C0.clone().to(DEV)
    Wd = projs[0][0]
    _launch_gemm(X.to(DEV), [_group(Wd, C, Ns[0], Wd.stride(0))], nf4=False, accumulate=True)
    _check_gemm(C, C0.float() + _ref_mm(X, Wd.cpu()), dtype, K, "gemm256 accumulate")


# ---------------------------------------------------------------- LoRA gradient products (csrc/lora_side.hip)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,R", [(1000, 264, 8), (512, 256, 16), (2048, 4096, 16), (777, 1032, 40), (3, 8, 1),
                                   (8192, 1024, 16)])
def test_lora_tn_matches_fp64(dtype, M, N, R):
    """G = s * P^T @ Z with P rounded to the activation dtype (where the reference holds a bf16 tensor),
    fp32 accumulate. Oracle: the same product in float64 on the CPU. Both output layouts, several problems in one
    launch, ragged M / N / rank, run-to-run bitwise determinism."""
    from unsloth_amd.kernels.utils import lora_tn
    Rp = (R + 3) // 4 * 4
    P = torch.randn(M, Rp + 4, generator=g(201))                    # ldp != R; second problem starts at column 4
    Z = torch.randn(M, N, generator=g(202)).to(dtype)
    Z2 = torch.randn(M, 2 * N, generator=g(203)).to(dtype)
    Pd, Zd, Z2d = P.to(DEV), Z.to(DEV), Z2.to(DEV)
    probs = [(Pd[:, :R], Zd, R, False, 0.5), (Pd[:, :R], Zd, R, True, 2.0), (Pd[:, 4:4 + R], Z2d, R, False, 1.0)]
    outs = lora_tn(probs)
    Pr = P.to(dtype).double()
    want = [0.5 * Pr[:, :R].t() @ Z.double(), (2.0 * Pr[:, :R].t() @ Z.double()).t(), Pr[:, 4:4 + R].t() @ Z2.double()]
    for o, w in zip(outs, want):
        assert o.dtype == torch.float32 and tuple(o.shape) == tuple(w.shape)
        err = (o.double().cpu() - w).abs().max().item()
        scale = w.abs().max().item() + 1e-30
        assert err <= 2e-5 * scale * max(1.0, (M / 512) ** 0.5), (err, scale)
    outs2 = lora_tn(probs)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,K,Ns,nf4", [(300, 512, (512, 128, 128), False), (4096, 3584, (3584, 512, 512), True),
                                        (64, 256, (264,), True), (2048, 1024, (1024,), False)])
def test_grouped_gemm_epilogue_bias(M, K, Ns, nf4):
    """uamd_gemm_group.bias: Y_g = X W_g^T + b_g (+ LoRA), the bias added in fp32 before the single rounding -- every
    kernel family (128 x 128 register-staged, fused NF4, 256 x 256 / 128 x 256 LDS-DMA, persistent walk) shares the
    epilogue. Qwen2's q/k/v shape (3584 -> 3584 | 512 | 512) included."""
    from unsloth_amd.kernels.utils import lora_linear_forward
    from unsloth_amd.nf4 import quantize_nf4
    X = (torch.randn(M, K, generator=g(31)) * 0.5).to(torch.bfloat16).to(DEV)
    projs, want = [], []
    for i, N in enumerate(Ns):
        W = (torch.randn(N, K, generator=g(32 + i)) * 0.05).to(torch.bfloat16)
        b = (torch.randn(N, generator=g(40 + i)) * 2.0).to(torch.bfloat16)
        A = (torch.randn(16, K, generator=g(50 + i)) * 0.05)
        B = (torch.randn(N, 16, generator=g(60 + i)) * 0.05)
        if nf4:
            packed, qs = quantize_nf4(W.to(DEV), compress_statistics=True)
            Wd = R.nf4_dequantize_state(packed, qs).float()
            projs.append((packed, qs, A.to(DEV), B.to(DEV), 2.0, b.to(DEV)))
        else:
            Wd = W.float()
            projs.append((W.to(DEV), None, A.to(DEV), B.to(DEV), 2.0, b.to(DEV)))
        xa = (X.float().cpu() @ A.to(torch.bfloat16).float().t()).to(torch.bfloat16).float()       # utils.py:1166 rounding point
        want.append(X.float().cpu() @ Wd.t() + b.float() + 2.0 * xa @ B.to(torch.bfloat16).float().t())
    outs = lora_linear_forward(X, projs)
    plain = lora_linear_forward(X, [p[:5] for p in projs])
    for o, p_, w, pr in zip(outs, plain, want, projs):
        assert rel_fro(o, w) < 4e-3, rel_fro(o, w)
        # against the bias-free launch: exactly the bias, up to the one rounding
        d = (o.float() - p_.float()).cpu() - pr[5].float().cpu()
        assert d.abs().max().item() <= 2.0 ** -7 * (w.abs().max().item())
