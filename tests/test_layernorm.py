"""LayerNorm (the vision towers' norm). CPU: the oracle's restatement against the fixture produced by the
REFERENCE's own Triton kernels (oracle/make_golden_layernorm.py, TRITON_INTERPRET=1). GPU: the HIP kernels against that
fixture, against the oracle in bf16 and against torch.nn.LayerNorm; in-place contract of the backward; patch_layernorm."""
import os

import pytest
import torch

from oracle import ref_ops as R
from tests._util import assert_ulp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_triton_layernorm.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=False)


CASES = ("f32_small", "f16_small", "f32_vit", "f16_ragged")


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_triton_fixture(gold, name):
    c = gold[name]
    dt = c["X"].dtype
    Y, r, mu = R.layernorm_forward(c["X"], c["W"], c["b"], c["eps"])
    dX = R.layernorm_backward(c["dY"], c["X"], c["W"], r, mu)
    if dt == torch.float32:
        torch.testing.assert_close(Y, c["Y"], rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(dX, c["dX"], rtol=2e-5, atol=2e-6)
    else:                       # fp16: fp32 arithmetic, one rounding -- equal up to the last fp32 bit before the rounding
        assert_ulp(Y, c["Y"], dt, ulps=1.0, what=name + " Y", allow_frac=1e-3)
        assert_ulp(dX, c["dX"], dt, ulps=1.0, what=name + " dX", allow_frac=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_layernorm_matches_reference_triton_fixture(gold, name):
    from unsloth_amd.kernels.layernorm import Fast_Layernorm
    c = gold[name]
    dt = c["X"].dtype
    X = c["X"].cuda().requires_grad_(True)
    dY = c["dY"].cuda().clone()
    Y = Fast_Layernorm.apply(X, c["W"].cuda(), c["b"].cuda(), c["eps"])
    Y.backward(dY)
    if dt == torch.float32:
        torch.testing.assert_close(Y.detach().cpu(), c["Y"], rtol=3e-6, atol=3e-6)
        torch.testing.assert_close(X.grad.cpu(), c["dX"], rtol=3e-5, atol=3e-6)
    else:
        assert_ulp(Y.detach().cpu(), c["Y"], dt, ulps=1.0, what=name + " Y", allow_frac=2e-3)
        assert_ulp(X.grad.cpu(), c["dX"], dt, ulps=1.0, what=name + " dX", allow_frac=2e-3)
    assert X.grad.data_ptr() == dY.data_ptr(), "dX must be written over dY (layernorm.py:104)"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,dim", [(64, 1280), (333, 1280), (17, 3584), (9, 5120), (5, 264), (3, 20000)])
def test_hip_layernorm_vs_oracle_and_torch(dtype, rows, dim):
    """bf16 / fp16 / fp32 at the ViT width (1280) and beyond, fp32 affine parameters too (mixed dtypes), odd widths through
    the generic kernel. The oracle carries the reference's rounding points; torch.nn.functional.layer_norm is the
    independent check (fp32 inside, one rounding)."""
    from unsloth_amd.kernels.layernorm import Fast_Layernorm
    g = torch.Generator().manual_seed(rows * 1000 + dim)
    X = (torch.randn(rows, dim, generator=g) * 2 + 0.5).to(dtype)
    dY = torch.randn(rows, dim, generator=g).to(dtype)
    for wdt in {dtype, torch.float32}:
        W, b = torch.rand(dim, generator=g).to(wdt), torch.rand(dim, generator=g).to(wdt)
        Yr, r, mu = R.layernorm_forward(X, W, b, 1e-6)
        dXr = R.layernorm_backward(dY, X, W, r, mu)
        Xg = X.cuda().requires_grad_(True)
        Y = Fast_Layernorm.apply(Xg.view(1, rows, dim), W.cuda(), b.cuda(), 1e-6)
        assert Y.shape == (1, rows, dim) and Y.dtype == dtype
        Y.backward(dY.cuda().view(1, rows, dim).clone())
        if dtype == torch.float32:
            torch.testing.assert_close(Y.detach().cpu()[0], Yr, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(Xg.grad.cpu(), dXr, rtol=1e-4, atol=1e-5)
        else:
            assert_ulp(Y.detach().cpu()[0], Yr, dtype, ulps=1.0, what="Y", allow_frac=2e-3)
            assert_ulp(Xg.grad.cpu(), dXr, dtype, ulps=1.0, what="dX", allow_frac=2e-3)
        Yt = torch.nn.functional.layer_norm(X.float(), (dim,), W.float(), b.float(), 1e-6)
        tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
        assert (Y.detach().float().cpu()[0] - Yt).abs().max().item() <= tol * (Yt.abs().max().item() + 1)


@pytest.mark.gpu
def test_patch_layernorm_routes_frozen_gpu_norms_only():
    from unsloth_amd.kernels import layernorm as L
    ln = torch.nn.LayerNorm(1280, eps=1e-6).cuda().to(torch.bfloat16)
    torch.nn.init.uniform_(ln.weight)
    torch.nn.init.uniform_(ln.bias)
    X = torch.randn(2, 40, 1280, device="cuda", dtype=torch.bfloat16)
    want = ln(X)
    calls = []
    real = L.Fast_Layernorm.apply
    L.patch_layernorm()
    try:
        L.Fast_Layernorm.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        assert ln(X).shape == want.shape and not calls          # trainable affine parameters: torch's own (dW / db needed)
        for p in ln.parameters():
            p.requires_grad_(False)
        got = ln(X)
        assert calls == [1]
        assert (got.float() - want.float()).abs().max().item() <= 2 ** -7 * want.float().abs().max().item()
        assert ln.cpu()(X.cpu()).shape == want.shape and calls == [1]        # host tensors: torch's own forward
    finally:
        L.Fast_Layernorm.apply = real
        L.unpatch_layernorm()
    assert torch.nn.LayerNorm.forward.__qualname__.startswith("LayerNorm")
