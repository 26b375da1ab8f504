"""-m gpu: the flash-attention kernels (csrc/attention.hip) against an fp32 softmax oracle (torch math on the
CPU): causal, GQA by head index, strided [B,T,H,D] views, ragged sequence lengths."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def ref_attention(q, k, v, scale, allowed=None):
    """q [B,T,Hq,D], k/v [B,T,Hk,D] fp32 -> (o [B,T,Hq,D], lse [B,Hq,T]); P rounded like the kernel is NOT modelled
    (the tolerance covers it). `allowed` [T,T] bool overrides the plain causal mask."""
    B, T, Hq, D = q.shape
    G = Hq // k.shape[2]
    kk = k.repeat_interleave(G, dim=2)
    vv = v.repeat_interleave(G, dim=2)
    s = torch.einsum("bthd,bshd->bhts", q, kk) * scale
    mask = torch.ones(T, T, dtype=torch.bool).tril() if allowed is None else allowed
    s = s.masked_fill(~mask, float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhts,bshd->bthd", p, vv)
    return o, lse


@pytest.fixture(params=[1, 2], ids=["one_block_per_item", "persistent_pingpong"])
def fwd_variant(request):
    """UAMD_TUNE_ATTN_VAR bit 0: attn_fwd_kernel always (one block per work item); bit 1: attn_fwd_ps_kernel always (one
    persistent workgroup per CU, ping-pong wave groups) -- the default picks by shape, here every shape runs on both."""
    from unsloth_amd import _lib
    L = _lib.lib()
    L.uamd_set_tuning(4, request.param)
    yield request.param
    L.uamd_set_tuning(4, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,Hq,Hk", [(1, 64, 4, 1), (2, 128, 8, 2), (1, 200, 4, 1), (1, 777, 8, 8), (2, 256, 8, 1),
                                       (1, 2048, 8, 2), (1, 31, 2, 1), (2, 1000, 4, 2), (1, 1500, 2, 2), (1, 640, 4, 2)])
def test_attn_forward_matches_fp32_oracle(fwd_variant, dtype, B, T, Hq, Hk):
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
    # norm-wise bounds at 2x the measured worst case over these shapes (tools/attn_err_probe.py on MI355X: relative
    # Frobenius error 2.0e-3 / 2.5e-4, worst single query row 3.2e-3 / 3.8e-4 for bf16 / fp16 -- i.e. the rounding of
    # the output and of P to the 16-bit type, nothing else)
    d = o.float().cpu() - o_ref
    assert (d.norm() / o_ref.norm()).item() <= (4e-3 if dtype == torch.bfloat16 else 5e-4)
    rows = d.flatten(0, -2).norm(dim=-1) / o_ref.flatten(0, -2).norm(dim=-1).clamp_min(1e-20)
    assert rows.max().item() <= (6.5e-3 if dtype == torch.bfloat16 else 8e-4)
    o2, _ = attn_forward(qd[..., :Hq * D].view(B, T, Hq, D), qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D),
                         qd[..., (Hq + Hk) * D:].view(B, T, Hk, D), scale)
    assert torch.equal(o, o2)


@pytest.fixture(params=[1, 2], ids=["fwd_one_block_per_item", "fwd_persistent"])
def bwd_variant(request):
    """The backward (dQ kernel + attn_bwd_dkdv4_kernel) behind either forward kernel (the LSE it reads comes from there)."""
    from unsloth_amd import _lib
    L = _lib.lib()
    L.uamd_set_tuning(4, request.param)
    yield request.param
    L.uamd_set_tuning(4, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,Hq,Hk", [(1, 64, 4, 1), (2, 128, 8, 2), (1, 200, 4, 1), (1, 333, 8, 8), (2, 256, 8, 1),
                                       (1, 1024, 8, 2), (1, 31, 2, 1), (1, 96, 4, 2), (1, 2048, 8, 2), (2, 777, 4, 4)])
def test_attn_backward_matches_fp32_autograd(bwd_variant, dtype, B, T, Hq, Hk):
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
        # measured (tools/attn_err_probe.py): relative Frobenius error <= 2.7e-3 (bf16) / 3.4e-4 (fp16) on every gradient
        rel = ((got.float().cpu() - want).norm() / want.norm()).item()
        assert rel <= (5.5e-3 if dtype == torch.bfloat16 else 7e-4), (name, rel)
    dq2, dk2, dv2 = attn_backward(do.to(DEV), q, k, v, o, lse, scale)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


def packed_mask(T, lengths, window):
    """dense allowed[q, key] of the reference's block-diagonal causal mask (utils/packing.py:650-693)."""
    allowed = torch.zeros(T, T, dtype=torch.bool)
    off = 0
    for n in list(lengths) + [T - sum(lengths)]:
        if n <= 0:
            continue
        blk = torch.ones(n, n, dtype=torch.bool).tril()
        if window is not None:
            idx = torch.arange(n)
            blk &= (idx[:, None] - idx[None, :]) < window
        allowed[off:off + n, off:off + n] = blk
        off += n
    return allowed


BAND_CASES = [  # T, Hq, Hk, packed lengths (None = one sequence), window
    (256, 4, 1, [100, 60, 96], None),
    (512, 8, 2, [1, 2, 3, 250, 64, 63, 129], None),       # document boundaries inside / on tile edges
    (384, 4, 4, None, 64),
    (777, 8, 2, None, 100),
    (1024, 8, 2, [300, 724], 128),
    (200, 2, 1, [33, 33, 33], 16),                          # trailing tokens form one more document
    (2048, 8, 2, [512, 1024, 512], None),
]


@pytest.mark.parametrize("T,Hq,Hk,lengths,window", BAND_CASES)
def test_attn_band_forward_backward(bwd_variant, T, Hq, Hk, lengths, window):
    """packed documents / sliding window: the kernels skip and mask by the (lo, hi) band; oracle = dense mask."""
    from unsloth_amd.kernels.attention import attention_band, attn_backward, attn_forward
    dtype, B, D = torch.bfloat16, 1, 128
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(5)).to(dtype)
    do = torch.randn(B, T, Hq, D, generator=g(6)).to(dtype)
    scale = 1.0 / math.sqrt(D)
    allowed = packed_mask(T, lengths or [T], window)
    qr = qkv[..., :Hq * D].view(B, T, Hq, D).float().requires_grad_(True)
    kr = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D).float().requires_grad_(True)
    vr = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D).float().requires_grad_(True)
    o_ref, lse_ref = ref_attention(qr, kr, vr, scale, allowed)
    o_ref.backward(do.float())
    band = attention_band(T, batch=B, seq_lengths=lengths, sliding_window=window, device=DEV)
    qd = qkv.to(DEV)
    q = qd[..., :Hq * D].view(B, T, Hq, D)
    k = qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qd[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    o, lse = attn_forward(q, k, v, scale, band)
    torch.testing.assert_close(lse.cpu(), lse_ref.detach(), rtol=1e-4, atol=2e-3)
    err = (o.float().cpu() - o_ref.detach()).abs().max().item()
    assert err <= 2e-2, err
    dq, dk, dv = attn_backward(do.to(DEV), q, k, v, o, lse, scale, band)
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        e = (got.float().cpu() - want).abs().max().item()
        ref = want.abs().max().item()
        assert e <= 3e-2 * max(ref, 1.0), (name, e, ref)
        assert torch.isfinite(got).all()


def test_attn_band_batch_window():
    """B > 1 with a sliding window (Mistral, mistral.py:116-120): same band for every row of the batch."""
    from unsloth_amd.kernels.attention import attention_band, flash_attention
    B, T, Hq, Hk, D, W = 2, 320, 4, 2, 128, 96
    q = torch.randn(B, T, Hq, D, generator=g(7)).to(torch.bfloat16)
    k = torch.randn(B, T, Hk, D, generator=g(8)).to(torch.bfloat16)
    v = torch.randn(B, T, Hk, D, generator=g(9)).to(torch.bfloat16)
    o_ref, _ = ref_attention(q.float(), k.float(), v.float(), 1.0 / math.sqrt(D), packed_mask(T, [T], W))
    o = flash_attention(q.to(DEV), k.to(DEV), v.to(DEV), None, attention_band(T, batch=B, sliding_window=W, device=DEV))
    assert (o.float().cpu() - o_ref).abs().max().item() <= 2e-2


def test_attn_forward_rescale_heavy_inputs(fwd_variant):
    """Scores that keep growing along the sequence (row max jumps by far more than the lazy-rescale threshold of 2^8 at
    many half tiles) and a constant-score case (no rescale after the first half tile)."""
    from unsloth_amd.kernels.attention import attn_forward
    B, T, Hq, Hk, D = 1, 1024, 4, 1, 128
    dtype = torch.bfloat16
    q = torch.randn(B, T, Hq, D, generator=g(21)).to(dtype)
    k = torch.randn(B, T, Hk, D, generator=g(22))
    k = (k * torch.linspace(0.2, 6.0, T)[None, :, None, None]).to(dtype)       # later keys score much higher
    v = torch.randn(B, T, Hk, D, generator=g(23)).to(dtype)
    scale = 1.0 / math.sqrt(D)
    for kk in (k, torch.zeros_like(k)):
        o_ref, lse_ref = ref_attention(q.float(), kk.float(), v.float(), scale)
        o, lse = attn_forward(q.to(DEV), kk.to(DEV), v.to(DEV), scale)
        torch.testing.assert_close(lse.cpu(), lse_ref, rtol=1e-4, atol=5e-3)
        assert (o.float().cpu() - o_ref).abs().max().item() <= 3e-2


PADDED_CASES = [  # B, T, Hq, Hk, D, packed lengths, window
    (1, 200, 7, 1, 128, None, None),                  # G = 7 (native: passes of 4 + 2 + 1 heads)
    (2, 333, 28, 4, 128, None, None),                 # Qwen2.5-7B / Qwen2-VL-7B head layout
    (1, 256, 6, 2, 128, [100, 156], None),            # G = 3 -> 4, packed
    (1, 192, 5, 1, 128, None, 48),                    # G = 5 -> 8, sliding window
    (2, 160, 8, 2, 64, None, None),                   # TinyLlama / Llama-3.2-1B head_dim
    (1, 300, 4, 4, 64, [64, 36, 200], None),
    (1, 128, 14, 2, 64, None, None),                  # Qwen2.5-0.5B: G = 7 AND head_dim 64
    (1, 96, 4, 2, 96, None, None),
    (1, 80, 4, 2, 36, None, None),                    # not a multiple of 8: the one shape class that still runs zero-padded
]


@pytest.mark.parametrize("B,T,Hq,Hk,D,lengths,window", PADDED_CASES)
def test_zero_padded_shapes_forward_backward(B, T, Hq, Hk, D, lengths, window):
    """Head dims below 128 and group sizes 3 / 5 / 6 / 7 are native since round 6 (zero-padded through copies before; a head dim
    that is not a multiple of 8 still is, kernels/attention.py _pad_qkv): outputs, LSE and all three gradients against the fp32
    oracle on the UNPADDED problem, and the gradient buffers keep the dQ | dK | dV column-block layout."""
    from unsloth_amd.kernels.attention import attention_band, attn_backward, attn_forward, native, supported
    dtype = torch.bfloat16
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(31)).to(dtype)
    do = torch.randn(B, T, Hq, D, generator=g(32)).to(dtype)
    scale = 1.0 / math.sqrt(D)
    allowed = packed_mask(T, lengths or [T], window) if (lengths or window) else None
    qr = qkv[..., :Hq * D].view(B, T, Hq, D).float().requires_grad_(True)
    kr = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D).float().requires_grad_(True)
    vr = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D).float().requires_grad_(True)
    o_ref, lse_ref = ref_attention(qr, kr, vr, scale, allowed)
    o_ref.backward(do.float())
    band = None
    if lengths or window:
        # the same documents in every batch row
        band = attention_band(T, batch=B, seq_lengths=(lengths + [T - sum(lengths)]) * B if lengths else None,
                              sliding_window=window, device=DEV)
    qd = qkv.to(DEV)
    q = qd[..., :Hq * D].view(B, T, Hq, D)
    k = qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qd[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    assert supported(q, k, v) and native(q, k, v) == (D % 8 == 0)
    o, lse = attn_forward(q, k, v, None, band)                      # default scale = 1 / sqrt(D) of the REAL head dim
    assert o.shape == (B, T, Hq, D) and lse.shape == (B, Hq, T)
    torch.testing.assert_close(lse.cpu(), lse_ref.detach(), rtol=1e-4, atol=2e-3)
    assert (o.float().cpu() - o_ref.detach()).abs().max().item() <= 2e-2
    dq, dk, dv = attn_backward(do.to(DEV), q, k, v, o, lse, None, band)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
    assert dk.data_ptr() == dq.data_ptr() + Hq * D * 2 and dv.data_ptr() == dk.data_ptr() + Hk * D * 2
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        e = (got.float().cpu() - want).abs().max().item()
        ref = want.abs().max().item()
        assert e <= 3e-2 * max(ref, 1.0), (name, e, ref)


SMALL_HEAD_CASES = [  # B, T, Hq, Hk, D, packed lengths, window, causal, dtype
    (1, 700, 32, 4, 64, None, None, True, torch.bfloat16),             # TinyLlama-1.1B (BASELINE config 1) head layout
    (2, 333, 8, 2, 64, None, None, True, torch.float16),
    (1, 1000, 16, 16, 80, [300, 700], None, False, torch.bfloat16),    # Qwen2-VL's vision tower: non-causal windows, head_dim 80
    (1, 640, 4, 2, 96, [100, 250, 290], None, True, torch.bfloat16),   # packed documents
    (1, 512, 14, 2, 64, None, 100, True, torch.bfloat16),              # Qwen2.5-0.5B: G = 7 AND head_dim 64, sliding window
    (1, 200, 2, 1, 16, None, None, True, torch.bfloat16), (1, 300, 4, 4, 112, None, None, True, torch.bfloat16),
]


@pytest.mark.parametrize("B,T,Hq,Hk,D,lengths,window,causal,dtype", SMALL_HEAD_CASES)
def test_head_dims_below_128_native_equal_the_zero_padded_run(B, T, Hq, Hk, D, lengths, window, causal, dtype):
    """Round 6: head dims below 128 (multiples of 8) without copies -- the kernels keep their 256-byte-row tiling but read nothing
    past a head's D elements as data: LDS-DMA lanes past D re-read slot 0 of their row, one side of every product holds zero
    registers (or zeroed LDS) there, rows past D of O^T / dQ^T / dK^T / dV^T are not stored. Against the SAME kernels on the
    zero-padded copies (rounds 2-5's path): the real columns go through the same arithmetic in the same order, so O, LSE, dQ,
    dK and dV are BIT-IDENTICAL. Inputs sit in one fused buffer so that a read past a head lands in its NEIGHBOUR's data (not
    zeros), and the buffers are poisoned behind their last element."""
    from unsloth_amd.kernels import attention as A
    G = Hq // Hk
    W = (Hq + 2 * Hk) * D
    pool = torch.full((B * T * W + 4096,), float("nan"), dtype=dtype, device=DEV)          # NaN behind the last row
    qkv = pool[:B * T * W].view(B, T, W)
    qkv.copy_((torch.randn(B, T, W, generator=g(51)) * 0.7).to(dtype))
    dpool = torch.full((B * T * Hq * D + 4096,), float("nan"), dtype=dtype, device=DEV)
    do = dpool[:B * T * Hq * D].view(B, T, Hq, D)
    do.copy_(torch.randn(B, T, Hq, D, generator=g(52)).to(dtype))
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = None
    if not causal:
        band = A.document_band(T, batch=B, seq_lengths=(lengths + [T - sum(lengths)]) * B if lengths else None, device=DEV)
    elif lengths or window:
        band = A.attention_band(T, batch=B, seq_lengths=(lengths + [T - sum(lengths)]) * B if lengths else None,
                                sliding_window=window, device=DEV)
    assert A.native(q, k, v)
    scale = 1.0 / math.sqrt(D)
    extra = () if causal else (False,)
    o, lse = A._forward_native(q, k, v, scale, band, *extra)
    assert o.shape == (B, T, Hq, D) and torch.isfinite(o.float()).all()
    dq, dk, dv = A._backward_native(do, q, k, v, o, lse, scale, band, *extra)
    for t in (dq, dk, dv):
        assert torch.isfinite(t.float()).all()
    qp, kp, vp = (A._pad_heads(t, 1, 1, D) for t in (q, k, v))
    op_, lsep = A._forward_native(qp, kp, vp, scale, band, *extra)
    assert torch.equal(o, op_[..., :D]) and torch.equal(lse, lsep)
    Tp = (T + 31) // 32 * 32
    lse_store = torch.as_strided(lsep, (B, Hq, Tp), (Hq * Tp, Tp, 1))
    dqp, dkp, dvp = A._backward_native(A._pad_heads(do, 1, 1, D), qp, kp, vp, op_, lse_store[:, :, :T], scale, band, *extra)
    assert torch.equal(dq, dqp[..., :D]) and torch.equal(dk, dkp[..., :D]) and torch.equal(dv, dvp[..., :D])
    # (the oracle: test_zero_padded_shapes_forward_backward and the non-causal document tests run the same shapes against fp32)


ODD_GROUP_CASES = [  # B, T, Hq, Hk, packed lengths, window, causal, dtype
    (2, 777, 14, 2, None, None, True, torch.bfloat16),             # G = 7: dK/dV passes of 4 + 2 + 1 heads
    (1, 2048, 28, 4, None, None, True, torch.bfloat16),            # Qwen2.5-7B / Qwen2-VL-7B text tower, plain causal (persistent forward)
    (1, 1024, 6, 2, [300, 200, 524], None, True, torch.bfloat16),  # G = 3 (2 + 1), packed documents
    (2, 640, 5, 1, None, 200, True, torch.float16),                # G = 5 (4 + 1), sliding window, fp16
    (1, 896, 12, 2, None, None, True, torch.bfloat16),             # G = 6 (4 + 2)
    (1, 900, 7, 1, [400, 500], None, False, torch.bfloat16),       # G = 7, non-causal inside documents
]


@pytest.mark.parametrize("B,T,Hq,Hk,lengths,window,causal,dtype", ODD_GROUP_CASES)
def test_group_sizes_3_5_6_7_native_equal_the_zero_padded_run(B, T, Hq, Hk, lengths, window, causal, dtype):
    """Round 6: group sizes 3 / 5 / 6 / 7 without copies -- the forward and dQ kernels run a KV head's query heads as virtual KV
    heads of 4 / 2 / 1 heads over the same K / V head, the dK / dV kernel in passes of 4, 2 and 1 heads. Against the SAME kernels
    on the zero-padded problem (all-zero dummy query heads up to 8 per KV head: rounds 2-5's path): a query row meets the same
    key tiles in the same order either way, so O, LSE and dQ are BIT-IDENTICAL; dK / dV sum the heads in a different order
    (norm-wise 2e-3). And against the fp32 oracle like every other shape."""
    from unsloth_amd.kernels import attention as A
    D, G = 128, Hq // Hk
    qkv = (torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(41)) * 0.7).to(dtype).to(DEV)
    do = torch.randn(B, T, Hq, D, generator=g(42)).to(dtype).to(DEV)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = None
    if not causal:
        band = A.document_band(T, batch=B, seq_lengths=(lengths + [T - sum(lengths)]) * B if lengths else None, device=DEV)
    elif lengths or window:
        band = A.attention_band(T, batch=B, seq_lengths=(lengths + [T - sum(lengths)]) * B if lengths else None,
                                sliding_window=window, device=DEV)
    assert A.native(q, k, v)
    scale = 1.0 / math.sqrt(D)
    extra = () if causal else (False,)
    o, lse = A._forward_native(q, k, v, scale, band, *extra)
    dq, dk, dv = A._backward_native(do, q, k, v, o, lse, scale, band, *extra)
    # the zero-padded problem on the same kernels
    qp = A._pad_heads(q, G, 8, D)
    op_, lsep = A._forward_native(qp, k, v, scale, band, *extra)
    o_cut = op_.view(B, T, Hk, 8, D)[:, :, :, :G].reshape(B, T, Hq, D)
    lse_cut = lsep.reshape(B, Hk, 8, T)[:, :, :G].reshape(B, Hq, T)
    assert torch.equal(o, o_cut) and torch.equal(lse, lse_cut)
    dop = A._pad_heads(do, G, 8, D)
    Tp = (T + 31) // 32 * 32
    lse_store = torch.as_strided(lsep, (B, Hk * 8, Tp), (Hk * 8 * Tp, Tp, 1))
    dqp, dkp, dvp = A._backward_native(dop, qp, k, v, op_, lse_store[:, :, :T], scale, band, *extra)
    assert torch.equal(dq, dqp.view(B, T, Hk, 8, D)[:, :, :, :G].reshape(B, T, Hq, D))
    for name, got, want in (("dk", dk, dkp), ("dv", dv, dvp)):
        err = (got.float() - want.float()).norm().item() / max(want.float().norm().item(), 1e-6)
        assert err <= 2e-3, (name, err)
    # and the oracle (fp32 softmax on the host) at a size it finishes in seconds
    if T <= 1024:
        if not causal:
            allowed = _doc_mask(T, lengths or [T])
        else:
            allowed = packed_mask(T, lengths or [T], window) if (lengths or window) else None
        qr, kr, vr = (t.float().cpu().requires_grad_(True) for t in (q, k, v))
        o_ref, lse_ref = ref_attention(qr, kr, vr, scale, allowed)
        o_ref.backward(do.float().cpu())
        torch.testing.assert_close(lse.cpu(), lse_ref.detach(), rtol=1e-4, atol=2e-3)
        assert (o.float().cpu() - o_ref.detach()).abs().max().item() <= 2e-2
        for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
            e = (got.float().cpu() - want).abs().max().item()
            assert e <= 3e-2 * max(want.abs().max().item(), 1.0), (name, e)


# ----------------------------------------------------------------------------------------------------------------------
# Non-causal attention inside documents (round 4): the vision tower of BASELINE config 4 (Qwen2-VL's ViT: every patch of an
# image attends every patch of the same image, `cu_seqlens` windows; reference: run_attention, attention_dispatch.py:298-617).
def _doc_mask(T, lengths):
    doc = torch.repeat_interleave(torch.arange(len(lengths)), torch.tensor(lengths))
    doc = torch.cat([doc, torch.full((T - doc.numel(),), len(lengths))]) if doc.numel() < T else doc[:T]
    return doc[:, None] == doc[None, :]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,Hq,Hk,D,lengths", [
    (1, 256, 4, 4, 128, None), (2, 200, 8, 2, 128, None), (1, 1024, 16, 16, 80, None),          # one document per row; the ViT's head_dim 80
    (1, 640, 4, 4, 128, [100, 37, 300, 203]), (1, 777, 2, 1, 128, [64, 64, 500]),               # windows, ragged tail document
    (1, 4096, 16, 16, 80, [1024, 3072])])
def test_noncausal_document_attention_forward_and_backward(dtype, B, T, Hq, Hk, D, lengths):
    from unsloth_amd.kernels.attention import attn_backward, attn_forward, document_band
    qkv = (torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(11)) * 1.0).to(dtype)
    do = torch.randn(B, T, Hq, D, generator=g(12)).to(dtype)
    scale = 1.0 / math.sqrt(D)
    qr = qkv[..., :Hq * D].view(B, T, Hq, D).float().requires_grad_(True)
    kr = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D).float().requires_grad_(True)
    vr = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D).float().requires_grad_(True)
    allowed = torch.ones(T, T, dtype=torch.bool) if lengths is None else _doc_mask(T, lengths)
    o_ref, lse_ref = ref_attention(qr, kr, vr, scale, allowed)
    o_ref.backward(do.float())
    qd = qkv.to(DEV)
    q = qd[..., :Hq * D].view(B, T, Hq, D)
    k = qd[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qd[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = None if lengths is None else document_band(T, batch=B, seq_lengths=lengths, device=DEV)
    o, lse = attn_forward(q, k, v, scale, band, causal=False)
    torch.testing.assert_close(lse.cpu(), lse_ref, rtol=1e-4, atol=2e-3)
    d = o.float().cpu() - o_ref.detach()
    assert (d.norm() / o_ref.norm()).item() <= (4e-3 if dtype == torch.bfloat16 else 5e-4)
    assert d.abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 4e-3)
    dq, dk, dv = attn_backward(do.to(DEV), q, k, v, o, lse, scale, band, causal=False)
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        err = (got.float().cpu() - want).norm() / want.norm()
        assert err <= (6e-3 if dtype == torch.bfloat16 else 8e-4), (name, float(err))
        assert (got.float().cpu() - want).abs().max().item() <= (3e-2 if dtype == torch.bfloat16 else 6e-3) * max(want.abs().max().item(), 1.0), name


def test_document_band_is_the_cu_seqlens_window():
    from unsloth_amd.kernels.attention import document_band
    lo, hi = document_band(10, batch=1, seq_lengths=[3, 5])
    assert lo.tolist() == [[0, 0, 0, 3, 3, 3, 3, 3, 8, 8]] and hi.tolist() == [[2, 2, 2, 7, 7, 7, 7, 7, 9, 9]]
    lo, hi = document_band(4, batch=2)
    assert lo.tolist() == [[0] * 4] * 2 and hi.tolist() == [[3] * 4] * 2


KD4_CASES = [  # (B, T, Hq, Hk, dtype, document lengths or None, causal)
    (1, 1024, 32, 8, torch.bfloat16, None, True), (1, 1024, 32, 8, torch.float16, None, True),
    (2, 2048, 32, 8, torch.bfloat16, None, True), (1, 1000, 32, 8, torch.bfloat16, None, True),
    (2, 777, 8, 8, torch.bfloat16, None, True), (1, 2048, 16, 8, torch.bfloat16, None, True),
    (1, 1024, 64, 8, torch.bfloat16, None, True), (1, 4096, 32, 8, torch.bfloat16, [700, 1348, 64, 33, 1951], True),
    (1, 2048, 16, 16, torch.float16, [1024, 1000, 24], False),
]


@pytest.mark.parametrize("B,T,Hq,Hk,dtype,lengths,causal", KD4_CASES)
def test_dkdv_generated_step_loops_are_bit_identical_to_the_cxx_body(B, T, Hq, Hk, dtype, lengths, causal):
    """attn_bwd_dkdv4_kernel: the steps over whole tiles run in the generated asm loops (csrc/attn_kd4_loop.inc, plain and
    masked); UAMD_TUNE_ATTN_VAR bit 2 sends EVERY step through the C++ body of the ragged tiles. Same arithmetic in the same
    order (Delta rides in the dP MFMAs' C operand in both): dQ, dK, dV bit-identical -- plain causal at G = 1, 2, 4, 8, ragged T,
    packed documents (band edges inside tiles), non-causal windows, both dtypes."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels import attention as A
    L = _lib.lib()
    qkv = (torch.randn(B, T, (Hq + 2 * Hk) * 128, generator=g(T + Hq)) * 0.7).to(dtype).to(DEV)
    q = qkv[..., :Hq * 128].view(B, T, Hq, 128)
    k = qkv[..., Hq * 128:(Hq + Hk) * 128].view(B, T, Hk, 128)
    v = qkv[..., (Hq + Hk) * 128:].view(B, T, Hk, 128)
    band = None
    if lengths is not None:
        band = (A.attention_band if causal else A.document_band)(T, batch=B, seq_lengths=lengths, device=DEV)
    o, lse = A.attn_forward(q, k, v, None, band, causal)
    do = torch.randn(o.shape, generator=g(5)).to(dtype).to(DEV)
    try:
        got = [t.clone() for t in A.attn_backward(do, q, k, v, o, lse, None, band, causal)]
        assert L.uamd_set_tuning(4, 4) == 0
        want = [t.clone() for t in A.attn_backward(do, q, k, v, o, lse, None, band, causal)]
    finally:
        L.uamd_set_tuning(4, 0)
    for name, a, b in zip(("dq", "dk", "dv"), got, want):
        assert not torch.isnan(a.float()).any(), name
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))


DYN_CASES = [  # (B, T, Hq, Hk, dtype, document lengths, sliding window)
    (1, 8192, 32, 8, torch.bfloat16, [1500, 64, 3, 700, 2048, 129, 511, 33, 1024, 900, 257, 64, 64, 600], None),   # 4 items per CU
    (2, 4096, 32, 8, torch.bfloat16, [4096, 100, 3000, 996], None),                   # a row that is one document + a ragged row
    (1, 8192, 28, 4, torch.bfloat16, [2000, 6000, 192], None),                        # G = 7: virtual KV heads of 4 / 2 / 1 heads
    (1, 4096, 32, 8, torch.float16, None, 512),                                       # sliding window only
    (1, 1000, 8, 2, torch.bfloat16, [64, 1, 63, 300, 572], None),                     # small grid (forced), single-tile items
    (1, 192, 4, 1, torch.bfloat16, [70, 122], None),                                  # fewer items than CUs (forced): every first claim fails
]


@pytest.mark.parametrize("B,T,Hq,Hk,dtype,lengths,window", DYN_CASES)
def test_persistent_forward_claims_items_bit_identical_to_the_static_deal(B, T, Hq, Hk, dtype, lengths, window):
    """attn_fwd_ps_kernel<BAND, DYN> (opt-in, UAMD_TUNE_ATTN_VAR bit 1): packed / windowed batches on the persistent kernel with items
    CLAIMED from a counter (every workgroup holds one claimed item ahead; the hand-off rides in a free ring stage). Which
    workgroup computes an item does not change its arithmetic: O and LSE bit-identical to the persistent kernel with the static
    deal (UAMD_TUNE_ATTN_VAR bit 3), run after run; against attn_fwd_kernel (one block per item: different phase order) within
    rounding; against the fp32 oracle on a slice."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels import attention as A
    L = _lib.lib()
    D = 128
    qkv = (torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(T + Hq)) * 0.8).to(dtype).to(DEV)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = A.attention_band(T, batch=B, seq_lengths=lengths, sliding_window=window, device=DEV)
    try:
        L.uamd_set_tuning(4, 2)                                   # persistent, claimed items
        runs = [tuple(t.clone() for t in A.attn_forward(q, k, v, None, band)) for _ in range(3)]
        L.uamd_set_tuning(4, 10)                                  # persistent, static deal
        o_s, lse_s = A.attn_forward(q, k, v, None, band)
        L.uamd_set_tuning(4, 1)                                   # one block per item
        o_1, lse_1 = A.attn_forward(q, k, v, None, band)
        L.uamd_set_tuning(4, 0)                                   # by shape
        o_d, lse_d = A.attn_forward(q, k, v, None, band)
    finally:
        L.uamd_set_tuning(4, 0)
    for o, lse in runs:
        assert not torch.isnan(o.float()).any()
        assert torch.equal(o, o_s) and torch.equal(lse, lse_s)
    assert torch.equal(o_d, o_1) and torch.equal(lse_d, lse_1)    # the default for packed / windowed batches: one block per item
    assert (o_1.float() - o_s.float()).abs().max().item() <= 2e-2
    torch.testing.assert_close(lse_1, lse_s, rtol=1e-4, atol=2e-3)
    # fp32 oracle on (row 0, KV head 0), first 1024 positions
    Tn, G = min(T, 1024), Hq // Hk
    allowed = packed_mask(T, ([n for n in lengths] if lengths else [T]), window)[:Tn, :Tn] if B == 1 else None
    if allowed is not None:
        o_ref, _ = ref_attention(q[:1, :Tn, :G].float().cpu(), k[:1, :Tn, :1].float().cpu(), v[:1, :Tn, :1].float().cpu(),
                                 1.0 / math.sqrt(D), allowed)
        assert (o_s[:1, :Tn, :G].float().cpu() - o_ref).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 4e-3)


def test_persistent_forward_counter_pairs_are_left_zeroed():
    """The claim counters come from a pool of 512 pairs handed out round-robin; a launch leaves its pair zeroed (the last
    workgroup to finish resets it). 1,200 launches walk the pool more than twice: every output equals the first."""
    from unsloth_amd import _lib
    from unsloth_amd.kernels import attention as A
    L = _lib.lib()
    B, T, Hq, Hk, D = 1, 640, 8, 2, 128
    qkv = torch.randn(B, T, (Hq + 2 * Hk) * D, generator=g(77)).to(torch.bfloat16).to(DEV)
    q = qkv[..., :Hq * D].view(B, T, Hq, D)
    k = qkv[..., Hq * D:(Hq + Hk) * D].view(B, T, Hk, D)
    v = qkv[..., (Hq + Hk) * D:].view(B, T, Hk, D)
    band = A.attention_band(T, batch=B, seq_lengths=[100, 300, 240], device=DEV)
    try:
        L.uamd_set_tuning(4, 2)
        o0, lse0 = A.attn_forward(q, k, v, None, band)
        bad = 0
        for i in range(1200):
            o, lse = A.attn_forward(q, k, v, None, band)
            if i % 100 == 99 or i > 1150:
                bad += int(not (torch.equal(o, o0) and torch.equal(lse, lse0)))
    finally:
        L.uamd_set_tuning(4, 0)
    assert bad == 0
