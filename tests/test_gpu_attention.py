"""-m gpu: the flash-attention kernels (csrc/attention.hip) against an fp32 softmax oracle (torch math on the
CPU): causal, GQA by head index, strided [B,T,H,D] views, ragged sequence lengths."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def ref_attention(q, k, v, scale):
    """q [B,T,Hq,D], k/v [B,T,Hk,D] fp32 -> (o [B,T,Hq,D], lse [B,Hq,T]); P rounded like the kernel is NOT modelled
    (the tolerance covers it)."""
    B, T, Hq, D = q.shape
    G = Hq // k.shape[2]
    kk = k.repeat_interleave(G, dim=2)
    vv = v.repeat_interleave(G, dim=2)
    s = torch.einsum("bthd,bshd->bhts", q, kk) * scale
    mask = torch.ones(T, T, dtype=torch.bool).tril()
    s = s.masked_fill(~mask, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhts,bshd->bthd", p, vv)
    return o, lse


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,Hq,Hk", [(1, 64, 4, 1), (2, 128, 8, 2), (1, 200, 4, 1), (1, 777, 8, 8), (2, 256, 8, 1),
                                       (1, 2048, 8, 2), (1, 31, 2, 1)])
def test_attn_forward_matches_fp32_oracle(dtype, B, T, Hq, Hk):
    from unsloth_amd.kernels.attention import attn_forward
    D = 128
    # Q/K/V as column slices of one fused QKV projection output [B*T, (Hq+2Hk)*D]: the strided layout the model uses
    qkv = (torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(1)) * 1.0).to(dtype)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    scale = 1.0 / math.sqrt(D)
    o_ref, lse_ref = ref_attention(q.float(), k.float(), v.float(), scale)
    qd = qkv.to(DEV)
    o, lse = attn_forward(qd[..., :Hq * D].view(B, T, Hq, D), qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D),
                          qd[..., (Hq + Hk) * D:].view(B, T, Hk, D), scale)
    torch.testing.assert_close(lse.cpu(), lse_ref, rtol=1e-4, atol=2e-3)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    err = (o.float().cpu() - o_ref).abs().max().item()
    assert err <= tol, err
    o2, _ = attn_forward(qd[..., :Hq * D].view(B, T, Hq, D), qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D),
                         qd[..., (Hq + Hk) * D:].view(B, T, Hk, D), scale)
    assert torch.equal(o, o2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,Hq,Hk", [(1, 64, 4, 1), (2, 128, 8, 2), (1, 200, 4, 1), (1, 333, 8, 8), (2, 256, 8, 1),
                                       (1, 1024, 8, 2), (1, 31, 2, 1), (1, 96, 4, 2)])
def test_attn_backward_matches_fp32_autograd(dtype, B, T, Hq, Hk):
    from unsloth_amd.kernels.attention import attn_backward, attn_forward
    D = 128
    qkv = (torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(2)) * 1.0).to(dtype)
    do = torch.randn(B, T, Hq, D, generator=g(3)).to(dtype)
    scale = 1.0 / math.sqrt(D)
    qr = qkv[..., :Hq * D].view(B, T, Hq, D).float().requires_grad_(True)
    kr = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D).float().requires_grad_(True)
    vr = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D).float().requires_grad_(True)
    o_ref, _ = ref_attention(qr, kr, vr, scale)
    o_ref.backward(do.float())
    qd = qkv.to(DEV)
    q = qd[..., :Hq * D].view(B, T, Hq, D)
    k = qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qd[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    o, lse = attn_forward(q, k, v, scale)
    dq, dk, dv = attn_backward(do.to(DEV), q, k, v, o, lse, scale)
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        err = (got.float().cpu() - want).abs().max().item()
        ref = want.abs().max().item()
        tol = (3e-2 if dtype == torch.bfloat16 else 6e-3) * max(ref, 1.0)
        assert err <= tol, (name, err, ref)
    dq2, dk2, dv2 = attn_backward(do.to(DEV), q, k, v, o, lse, scale)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
