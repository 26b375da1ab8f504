"""Shared helpers of the test-suite (not a test module)."""
import torch

EPS = {torch.float32: 2.0 ** -23, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


def assert_ulp(actual, expected, dtype, ulps=1.0, atol=None, what="", allow_frac=0.0):
    """|actual - expected| <= ulps * eps(dtype) * |expected| + atol elementwise. `allow_frac` tolerates a
    fraction of elements at 2x the bound (transcendental implementations differ in the last fp32 bit,
    which can flip a rounding to the 16-bit dtype)."""
    a = actual.detach().float().cpu()
    b = expected.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values in result"
    eps = EPS[dtype]
    if atol is None:
        atol = eps * float(b.abs().mean() + 1e-30) * 0.5
    bound = ulps * eps * b.abs() + atol
    err = (a - b).abs()
    bad = err > bound
    if bad.any():
        worse = err > 2 * bound
        frac = float(bad.float().mean())
        idx = torch.nonzero(bad)[0].tolist()
        msg = (f"{what}: {int(bad.sum())}/{bad.numel()} beyond {ulps} ulp ({frac:.2e}); first at {idx}: "
               f"got {a[tuple(idx)].item()!r} want {b[tuple(idx)].item()!r}; max err {err.max().item():.4e}")
        assert not worse.any() and frac <= allow_frac, msg


def rel_fro(actual, expected):
    a = actual.detach().float().cpu()
    b = expected.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
