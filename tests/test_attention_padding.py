"""CPU: the zero-padding algebra of kernels/attention.py (head dims below 128 on kernels built for head_dim 128; group sizes
3/5/6/7 are native since round 6 and only pass through) with the two kernel launches replaced by an fp32 torch emulation that has the SAME
contract as the HIP kernels (flash-style backward from the saved LSE, dQ|dK|dV column blocks of one buffer, LSE storage
padded to 32 positions) and REFUSES any non-native shape. What is checked is exactly what the wrappers add: the padded
problem's outputs / gradients, cut back, equal the unpadded problem's -- including that the all-zero dummy query heads
(whose LSE is passed as 0) contribute nothing to dK / dV. The kernels themselves: tests/test_gpu_attention.py."""
import math

import pytest
import torch

from unsloth_amd.kernels import attention as A


def _allowed(B, T, band):
    pos = torch.arange(T)
    al = (pos[None, :] <= pos[:, None])[None].expand(B, T, T)
    if band is not None:
        al = al & (pos[None, None, :] >= band[0][:, :, None].long())
    return al


def _scores(q, k, scale, band):
    B, T, Hq, D = q.shape
    G = Hq // k.shape[2]
    assert D == 128 and 1 <= G <= 8, "the emulated kernel takes native shapes only"
    s = torch.einsum("bthd,bshd->bhts", q.float(), k.float().repeat_interleave(G, dim=2)) * scale
    return s.masked_fill(~_allowed(B, T, band)[:, None], float("-inf")), G


def emu_forward(q, k, v, scale, band):
    B, T, Hq, D = q.shape
    s, G = _scores(q, k, scale, band)
    lse_t = torch.logsumexp(s, -1)                                           # [B,Hq,T]
    o = torch.einsum("bhts,bshd->bthd", torch.exp(s - lse_t[..., None]), v.float().repeat_interleave(G, dim=2))
    Tp = (T + 31) // 32 * 32
    lse = torch.zeros(B, Hq, Tp)
    lse[:, :, :T] = lse_t
    return o.to(q.dtype).contiguous(), lse[:, :, :T]


def emu_backward(do, q, k, v, o, lse, scale, band):
    B, T, Hq, D = q.shape
    Hk = k.shape[2]
    s, G = _scores(q, k, scale, band)
    p = torch.exp(s - lse[..., None])                                        # from the SAVED lse, like the kernels
    p = p.masked_fill(~_allowed(B, T, band)[:, None], 0.0)
    dof, vf = do.float(), v.float().repeat_interleave(G, dim=2)
    dv = torch.einsum("bhts,bthd->bshd", p, dof).view(B, T, Hk, G, D).sum(3)
    dp = torch.einsum("bthd,bshd->bhts", dof, vf)
    delta = (dof * o.float()).sum(-1).permute(0, 2, 1)                       # [B,Hq,T]
    ds = p * (dp - delta[..., None]) * scale
    dq = torch.einsum("bhts,bshd->bthd", ds, k.float().repeat_interleave(G, dim=2))
    dk = torch.einsum("bhts,bthd->bshd", ds, q.float()).view(B, T, Hk, G, D).sum(3)
    dqkv = torch.empty(B, T, (Hq + 2 * Hk) * D, dtype=q.dtype)
    dqkv[..., :Hq * D] = dq.reshape(B, T, Hq * D)
    dqkv[..., Hq * D:(Hq + Hk) * D] = dk.reshape(B, T, Hk * D)
    dqkv[..., (Hq + Hk) * D:] = dv.reshape(B, T, Hk * D)
    return (dqkv[..., :Hq * D].view(B, T, Hq, D), dqkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D),
            dqkv[..., (Hq + Hk) * D:].view(B, T, Hk, D))


@pytest.fixture
def emulated(monkeypatch):
    monkeypatch.setattr(A._lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(A, "native", lambda q, k, v: q.shape[-1] == 128 and (q.shape[2] // k.shape[2]) in A._GROUPS)
    monkeypatch.setattr(A, "supported", lambda q, k, v: q.shape[-1] <= 128 and q.shape[2] // k.shape[2] <= 8)
    monkeypatch.setattr(A, "_forward_native", emu_forward)
    monkeypatch.setattr(A, "_backward_native", emu_backward)


@pytest.mark.parametrize("B,T,Hq,Hk,D,lengths,window", [
    (1, 40, 7, 1, 128, None, None), (2, 33, 28, 4, 128, None, None), (1, 64, 6, 2, 128, [20, 44], None),
    (1, 48, 5, 1, 128, None, 12), (2, 32, 8, 2, 64, None, None), (1, 50, 4, 4, 64, [14, 6, 30], None),
    (1, 31, 14, 2, 64, None, None), (1, 24, 4, 2, 96, None, None), (1, 40, 3, 1, 40, None, None)])
def test_padding_wrappers_equal_the_unpadded_problem(emulated, B, T, Hq, Hk, D, lengths, window):
    gen = torch.Generator().manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, generator=gen)               # fp32 end to end: the algebra is exact
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    do = torch.randn(B, T, Hq, D, generator=gen)
    band = None
    if lengths or window:
        band = A.attention_band(T, batch=B, seq_lengths=(lengths * B) if lengths else None, sliding_window=window)
    # the unpadded truth, by autograd
    qr, kr, vr = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    G = Hq // Hk
    s = torch.einsum("bthd,bshd->bhts", qr, kr.repeat_interleave(G, dim=2)) / math.sqrt(D)
    s = s.masked_fill(~_allowed(B, T, band)[:, None], float("-inf"))
    o_ref = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vr.repeat_interleave(G, dim=2))
    o_ref.backward(do)
    o, lse = A.attn_forward(q, k, v, None, band)
    assert o.shape == (B, T, Hq, D) and lse.shape == (B, Hq, T) and lse.stride(1) == (T + 31) // 32 * 32
    torch.testing.assert_close(o, o_ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lse, torch.logsumexp(s, -1).detach(), rtol=1e-5, atol=1e-5)
    dq, dk, dv = A.attn_backward(do, q, k, v, o, lse, None, band)
    es = q.element_size()
    assert dk.data_ptr() == dq.data_ptr() + Hq * D * es and dv.data_ptr() == dk.data_ptr() + Hk * D * es
    torch.testing.assert_close(dq, qr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dk, kr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dv, vr.grad, rtol=1e-4, atol=1e-5)


def test_native_shapes_are_not_padded(emulated):
    q = torch.randn(1, 16, 8, 128)
    k = torch.randn(1, 16, 2, 128)
    calls = []
    real = A._pad_qkv
    A._pad_qkv = lambda *a: (calls.append(1), real(*a))[1]
    try:
        A.attn_forward(q, k, k.clone())
    finally:
        A._pad_qkv = real
    assert not calls
