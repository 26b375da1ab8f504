"""-m gpu: every value of a `uamd_set_tuning` knob that README's Switches table lists computes the same thing as the default.

The knobs pick between schedules of one computation (grid shape, raster order, cache hints, tile height), so the results must be
bit-identical to the default's -- except `UAMD_TUNE_RMS_VAR`, whose two kernels reduce a row in a different order (wave shuffle
tree vs. block LDS tree): fp32 sums within rounding, 16-bit outputs within 1 ulp. Knobs 3, 4, 6, 7, 8 and 9 (transposing dequant,
attention forward kernel, GEMM tile height / persistence / plain epilogue, x4 dequant) have their own parametrised tests in
tests/test_gpu_nf4_gemm.py and tests/test_gpu_attention.py, knob 10 (fused activation schedules) in tests/test_gpu_glu_fused.py; this
file covers 0, 1, 2 and 5.
(The reference has no such knobs: its Triton launches are fixed, kernels/swiglu.py:41-60, rms_layernorm.py:23-60.)"""
import pytest
import torch

from tests._util import assert_ulp

pytestmark = pytest.mark.gpu
DEV = "cuda"
GLU_VAR, GROUP_M, STREAM_NT, RMS_VAR = 0, 1, 2, 5
DEFAULT = {GLU_VAR: 2, GROUP_M: 8, STREAM_NT: 0, RMS_VAR: 1}


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture
def lib():
    from unsloth_amd import _lib
    L = _lib.lib()
    yield L
    for k, v in DEFAULT.items():
        L.uamd_set_tuning(k, v)


def _glu(n_rows, dtype):
    from unsloth_amd.kernels.swiglu import swiglu_fg_kernel, swiglu_DWf_DW_dfg_kernel
    e = torch.randn(n_rows, 14336, generator=g(1)).to(dtype).to(DEV)
    gt = torch.randn(n_rows, 14336, generator=g(2)).to(dtype).to(DEV)
    dw = torch.randn(n_rows, 14336, generator=g(3)).to(dtype).to(DEV)
    h = swiglu_fg_kernel(e, gt)
    outs = swiglu_DWf_DW_dfg_kernel(dw.clone(), e.clone(), gt.clone())
    return (h,) + tuple(outs)


def _rms(rows, dim, dtype):
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    X = torch.randn(rows, dim, generator=g(4)).to(dtype).to(DEV).requires_grad_(True)
    W = torch.rand(dim, generator=g(5)).to(dtype).to(DEV)
    dY = torch.randn(rows, dim, generator=g(6)).to(dtype).to(DEV)
    Y = Fast_RMS_Layernorm.apply(X, W, 1e-5, False)
    Y.backward(dY)
    return Y.detach(), X.grad


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows", [3, 700])                       # 700 x 14336 = past the 2048-block cap of variant 0
def test_glu_grid_variants_and_cache_hints_are_bitwise_equal(lib, rows, dtype):
    want = _glu(rows, dtype)
    for var in (0, 1, 2):
        for nt in (0, 1, 2, 3):
            assert lib.uamd_set_tuning(GLU_VAR, var) == 0 and lib.uamd_set_tuning(STREAM_NT, nt) == 0
            got = _glu(rows, dtype)
            for a, b in zip(want, got):
                assert torch.equal(a, b), (var, nt)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,dim", [(37, 4096), (5, 100), (9, 16384)])
def test_rmsnorm_kernel_variants_and_cache_hints(lib, rows, dim, dtype):
    want = _rms(rows, dim, dtype)
    for var in (0, 1):
        for nt in (0, 1, 2, 3):
            assert lib.uamd_set_tuning(RMS_VAR, var) == 0 and lib.uamd_set_tuning(STREAM_NT, nt) == 0
            got = _rms(rows, dim, dtype)
            for a, b, what in zip(want, got, ("Y", "dX")):
                if var == DEFAULT[RMS_VAR]:
                    assert torch.equal(a, b), (var, nt, what)
                else:
                    assert_ulp(b, a.float().cpu(), dtype, ulps=32 if dtype == torch.float32 else 1,
                               what=f"rms var {var} {what}", allow_frac=2e-3)


@pytest.mark.parametrize("group_m", [1, 2, 4, 8, 64])
def test_gemm256_raster_group_height_is_bitwise_neutral(lib, group_m):
    """knob 1 only reorders which tile a block takes (L2 reuse): 2048 x 4096 x 1024 = 8 x 16 tiles of the 256 x 256 kernel."""
    from unsloth_amd.kernels import utils as U
    from unsloth_amd.kernels.utils import lora_linear_forward
    X = torch.randn(2048, 1024, generator=g(7)).to(torch.bfloat16).to(DEV)
    W = (torch.randn(4096, 1024, generator=g(8)) * 0.05).to(torch.bfloat16).to(DEV)
    old = U.GEMM256_MODE
    U.GEMM256_MODE = "on"
    try:
        (want,) = lora_linear_forward(X, [(W, None, None, None, None)])
        assert lib.uamd_set_tuning(GROUP_M, group_m) == 0
        (got,) = lora_linear_forward(X, [(W, None, None, None, None)])
        assert torch.equal(want, got)
        ref = X.float() @ W.float().t()
        assert (got.float() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max()
    finally:
        U.GEMM256_MODE = old


def test_set_tuning_refuses_what_it_does_not_know(lib):
    assert lib.uamd_set_tuning(-1, 0) != 0 and lib.uamd_set_tuning(12, 0) != 0 and lib.uamd_set_tuning(0, -1) != 0
