"""-m gpu: the hot-path kernels at BASELINE.json's FULL sizes (Llama-3-8B: hidden 4096, intermediate 14336, vocab
128256, 32/8 heads of 128, 4 x 2048 tokens) through size-independent properties -- exact scaling laws, inverse
round trips, idempotence, partition independence, conservation sums -- plus spot checks against fp64 on sampled
entries and the empty-input edge of every entry point. The small-size tests compare element by element with the
oracle; these make sure nothing changes when the grids, strides and 32-bit index ranges are the real ones."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, H, I, V, HQ, HK, D = 8192, 4096, 14336, 128256, 32, 8, 128
BF = torch.bfloat16


def gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def test_rmsnorm_fullsize_scale_laws_and_row_statistics():
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    x = torch.randn(T, H, device=DEV, generator=gen(1)).to(BF)
    w = torch.rand(H, device=DEV, generator=gen(2)).to(BF)
    y = Fast_RMS_Layernorm.apply(x, w, 1e-5, False)
    assert torch.equal(y, Fast_RMS_Layernorm.apply(x, w, 1e-5, False))                 # deterministic
    # normalised rows have unit RMS (before the weight): check through w = 1
    y1 = Fast_RMS_Layernorm.apply(x, torch.ones_like(w), 1e-5, False).float()
    rms = y1.pow(2).mean(-1).sqrt()
    assert float((rms - 1).abs().max()) < 2 ** -7
    # homogeneity: scaling the input by 2^k only moves eps (1e-5 against mean x^2 ~ 1), which can tip each of the
    # two roundings (normed -> bf16, * W -> bf16) by one ulp
    y4 = Fast_RMS_Layernorm.apply(x * 4, w, 1e-5, False)
    d = (y4.float() - y.float()).abs()
    assert float((d > 2 ** -6 * y.float().abs() + 1e-6).float().mean()) == 0.0
    assert float((d > 0).float().mean()) < 1e-2
    # backward: dX is orthogonal to x per row for w = 1  (sum_j dX_j x_j = 0 analytically)
    dy = torch.randn(T, H, device=DEV, generator=gen(3)).to(BF)
    xg = x.clone().requires_grad_(True)
    Fast_RMS_Layernorm.apply(xg, torch.ones_like(w), 1e-5, False).backward(dy.clone())
    dot = (xg.grad.float() * x.float()).sum(-1)
    scale = (xg.grad.float().abs() * x.float().abs()).sum(-1)
    assert float((dot.abs() / scale).max()) < 2e-3


def test_rope_fullsize_backward_inverts_forward_and_preserves_norms():
    from unsloth_amd.kernels.rope_embedding import fast_rope_embedding
    from unsloth_amd.models.llama import RopeTables
    from transformers import LlamaConfig
    cfg = LlamaConfig(hidden_size=H, num_attention_heads=HQ, num_key_value_heads=HK, head_dim=D,
                      max_position_embeddings=8192, rope_parameters={"rope_type": "default", "rope_theta": 5e5})
    cos, sin = RopeTables(cfg).get(2048, torch.device(DEV), BF)
    B, S = 4, 2048
    qkv = torch.randn(B, S, (HQ + 2 * HK) * D, device=DEV, generator=gen(4)).to(BF)
    Q = qkv[..., :HQ * D].view(B, S, HQ, D).transpose(1, 2)              # strided views, like the model
    K = qkv[..., HQ * D:(HQ + HK) * D].view(B, S, HK, D).transpose(1, 2)
    q0, k0 = Q.clone(), K.clone()
    pos = torch.arange(S, dtype=torch.int32, device=DEV).unsqueeze(0).expand(B, S).contiguous()
    Qr, Kr = fast_rope_embedding(Q, K, cos, sin, pos)
    # pairwise norms (x_i, x_{i+64}) are preserved by a rotation: bf16 rounding only
    n0 = q0.float().pow(2).view(B, HQ, S, 2, D // 2).sum(3)
    n1 = Qr.float().pow(2).view(B, HQ, S, 2, D // 2).sum(3)
    assert float(((n1 - n0).abs() / (n0 + 1e-3)).max()) < 3e-2
    # position 0 is the identity (cos 1, sin 0)
    assert torch.equal(Qr[:, :, 0], q0[:, :, 0]) and torch.equal(Kr[:, :, 0], k0[:, :, 0])
    # the backward is the inverse rotation: two roundings away from the input
    from unsloth_amd.kernels.rope_embedding import _launch_qk
    Qb, Kb = Qr.clone(), Kr.clone()
    _launch_qk(Qb, Kb, cos, sin, pos, True)
    assert float((Qb.float() - q0.float()).abs().max()) <= 2 ** -6 * float(q0.float().abs().max())
    assert float((Kb.float() - k0.float()).abs().max()) <= 2 ** -6 * float(k0.float().abs().max())


def test_swiglu_fullsize_is_linear_in_the_gate_and_backward_matches_autograd_sample():
    from unsloth_amd.kernels.swiglu import swiglu_DWf_DW_dfg_kernel, swiglu_fg_kernel
    e = torch.randn(T, I, device=DEV, generator=gen(5)).to(BF)
    g = torch.randn(T, I, device=DEV, generator=gen(6)).to(BF)
    h = swiglu_fg_kernel(e, g)
    assert torch.equal(swiglu_fg_kernel(e, g * 2), h * 2)               # powers of two commute with the rounding
    assert torch.equal(swiglu_fg_kernel(e, -g), -h)
    assert torch.equal(swiglu_fg_kernel(torch.zeros_like(e), g), torch.zeros_like(h))
    # in-place backward, spot-checked against fp64 autograd on 4 rows spread over the range
    rows = torch.tensor([0, 2731, 5000, T - 1], device=DEV)
    DW = torch.randn(T, I, device=DEV, generator=gen(7)).to(BF)
    ed, gd, dwd = e[rows].double().requires_grad_(True), g[rows].double().requires_grad_(True), DW[rows].double()
    (torch.nn.functional.silu(ed) * gd * dwd).sum().backward()
    hh, df, de = swiglu_DWf_DW_dfg_kernel(DW.clone(), e.clone(), g.clone())
    assert torch.equal(hh, h)
    assert float((de[rows].double() - ed.grad).abs().max()) < 3e-2 * float(ed.grad.abs().max())
    # df = DW * f (what the reference returns in e's buffer, swiglu.py:112-125)
    f = torch.nn.functional.silu(e[rows].double())
    assert float((df[rows].double() - dwd * f).abs().max()) < 2e-2 * float((dwd * f).abs().max())


def test_cross_entropy_full_vocab_conservation():
    from unsloth_amd.kernels.cross_entropy_loss import Fast_CrossEntropyLoss
    n = 2048
    logits = (torch.randn(n, V, device=DEV, generator=gen(8)) * 4).to(BF)
    labels = torch.randint(0, V, (n,), device=DEV, generator=gen(9))
    labels[::10] = -100
    lg = logits.clone().requires_grad_(True)
    losses = Fast_CrossEntropyLoss.apply(lg, labels, 0, 0)
    assert bool((losses[::10] == 0).all()) and bool((losses >= 0).all())
    ref = torch.nn.functional.cross_entropy(logits[:64].float(), labels[:64], reduction="none", ignore_index=-100)
    assert float((losses[:64] - ref).abs().max()) < 2e-3
    up = torch.rand(n, device=DEV, generator=gen(10)) + 0.5
    (losses * up).sum().backward()
    gsum = lg.grad.float().sum(-1)                                       # softmax - onehot sums to zero per row
    assert float(gsum.abs().max()) < 2e-2
    assert bool((lg.grad[::10] == 0).all())                              # ignored rows: exactly zero gradient
    picked = lg.grad.float().gather(1, labels.clamp_min(0).unsqueeze(1)).squeeze(1)
    keep = labels != -100
    assert bool((picked[keep] <= 0).all())                               # p - 1 <= 0 at the label


def test_nf4_fullsize_idempotence_and_oracle_equality():
    from oracle.ref_ops import nf4_dequantize_state
    from unsloth_amd.nf4 import dequantize_nf4, quantize_nf4
    W = (torch.randn(I, H, device=DEV, generator=gen(11)) * 0.02).to(BF)          # gate_proj
    packed, qs = quantize_nf4(W, compress_statistics=False)
    deq = dequantize_nf4(packed, qs)
    assert tuple(deq.shape) == (I, H)
    # full 58.7 M parameters against the numpy restatement, bit for bit (a few seconds on the host)
    assert torch.equal(deq.cpu(), nf4_dequantize_state(packed, qs))
    # the transposed kernel is the same numbers
    assert torch.equal(dequantize_nf4(packed, qs, transpose=True), deq.t())
    # quantise(dequantise(q)) == q on codes and scales: the decoded max of every block is +-1 * absmax. Done in
    # fp32 so no second rounding enters.
    qs32 = type(qs)(absmax=qs.absmax, shape=qs.shape, code=qs.code, blocksize=qs.blocksize, quant_type="nf4",
                    dtype=torch.float32)
    deq32 = dequantize_nf4(packed, qs32)
    packed2, qs2 = quantize_nf4(deq32, compress_statistics=False)
    assert torch.equal(packed2, packed) and torch.equal(qs2.absmax, qs.absmax)
    # nested statistics round trip: double-quantised state decodes within the 8-bit map's resolution of absmax
    packed3, qs3 = quantize_nf4(W, compress_statistics=True)
    assert torch.equal(packed3, packed)
    from unsloth_amd.nf4 import absmax_f32
    rel = (absmax_f32(qs3) - qs.absmax).abs() / qs.absmax.abs().clamp_min(1e-12)
    assert float(rel.max()) < 0.1 and float(rel.mean()) < 0.01


def test_gemm_fullsize_partition_independence_and_fp64_samples():
    from unsloth_amd.kernels.utils import lora_linear_forward
    X = torch.randn(T, H, device=DEV, generator=gen(12)).to(BF)
    W = (torch.randn(I, H, device=DEV, generator=gen(13)) * 0.02).to(BF)
    Y = lora_linear_forward(X, [(W, None, None, None, None)])[0]
    assert tuple(Y.shape) == (T, I)
    # a tile's value does not depend on which other tiles are in the launch: row and column sub-problems agree
    Y_rows = lora_linear_forward(X[4096:4096 + 512], [(W, None, None, None, None)])[0]
    assert torch.equal(Y_rows, Y[4096:4096 + 512])
    Y_cols = lora_linear_forward(X, [(W[1024:1024 + 512], None, None, None, None)])[0]
    assert torch.equal(Y_cols, Y[:, 1024:1024 + 512])
    # sampled entries against fp64 dot products
    g = torch.Generator().manual_seed(0)
    r = torch.randint(0, T, (256,), generator=g).to(DEV)
    c = torch.randint(0, I, (256,), generator=g).to(DEV)
    want = (X[r].double() * W[c].double()).sum(-1)
    got = Y[r, c].double()
    assert float((got - want).abs().max()) <= 2 ** -8 * float(want.abs().max()) + 1e-3
    # exact scaling law: (2X) W^T == 2 (X W^T) in bf16 and fp32 accumulate
    assert torch.equal(lora_linear_forward(X[:1024] * 2, [(W, None, None, None, None)])[0], Y[:1024] * 2)


def test_attention_fullsize_rows_of_p_sum_to_one_and_causality():
    from unsloth_amd.kernels.attention import attn_backward, attn_forward
    B, S = 1, 2048
    qkv = torch.randn(B, S, (HQ + 2 * HK) * D, device=DEV, generator=gen(14)).to(BF)
    q = qkv[..., :HQ * D].view(B, S, HQ, D)
    k = qkv[..., HQ * D:(HQ + HK) * D].view(B, S, HK, D)
    v = qkv[..., (HQ + HK) * D:].view(B, S, HK, D)
    o, lse = attn_forward(q, k, v)
    # V = 1  =>  O = sum_j P_ij = 1
    ones = torch.ones_like(v)
    o1, _ = attn_forward(q, k, ones)
    assert float((o1.float() - 1).abs().max()) <= 2 ** -7
    # causality: keys / values after position t cannot change outputs up to t -- bit for bit
    k2, v2 = k.clone(), v.clone()
    k2[:, 1536:] = torch.randn_like(k2[:, 1536:])
    v2[:, 1536:] = torch.randn_like(v2[:, 1536:])
    o2, lse2 = attn_forward(q, k2, v2)
    assert torch.equal(o2[:, :1536], o[:, :1536]) and torch.equal(lse2[..., :1536], lse[..., :1536])
    # first position attends itself only: O[0] = V[0] of its KV head
    assert torch.equal(o[:, 0].view(B, HK, HQ // HK, D), v[:, 0].unsqueeze(2).expand(B, HK, HQ // HK, D))
    # backward conservation: sum_i dS_ij over keys is zero per query  =>  with dO = O-independent constant c * 1
    # and V = 1, dP - Delta = 0 everywhere: dQ = dK = 0 exactly, dV_j = sum_i P_ij c
    do = torch.full_like(o1, 0.5)
    dq, dk, dv = attn_backward(do, q, k, ones, o1, attn_forward(q, k, ones)[1])
    assert float(dq.float().abs().max()) < 2e-2 and float(dk.float().abs().max()) < 2e-2
    # total probability mass: sum_j dV_j[d] = 0.5 * (#queries * G) per kv head and channel
    tot = dv.float().sum(1)                                               # [B, HK, D]
    assert float((tot / (0.5 * S * (HQ // HK)) - 1).abs().max()) < 1e-2


def test_empty_inputs_are_no_ops():
    from unsloth_amd.kernels.attention import attn_forward
    from unsloth_amd.kernels.cross_entropy_loss import Fast_CrossEntropyLoss
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    from unsloth_amd.kernels.swiglu import swiglu_fg_kernel
    from unsloth_amd.kernels.utils import lora_linear_forward
    w = torch.ones(H, device=DEV, dtype=BF)
    assert Fast_RMS_Layernorm.apply(torch.empty(0, H, device=DEV, dtype=BF), w, 1e-5, False).shape == (0, H)
    assert swiglu_fg_kernel(torch.empty(0, I, device=DEV, dtype=BF), torch.empty(0, I, device=DEV, dtype=BF)).shape == (0, I)
    assert Fast_CrossEntropyLoss.apply(torch.empty(0, 1000, device=DEV, dtype=BF),
                                       torch.empty(0, dtype=torch.long, device=DEV), 0, 0).shape == (0,)
    W = torch.zeros(256, H, device=DEV, dtype=BF)
    assert lora_linear_forward(torch.empty(0, H, device=DEV, dtype=BF), [(W, None, None, None, None)])[0].shape == (0, 256)
    o, lse = attn_forward(torch.empty(1, 0, 8, D, device=DEV, dtype=BF), torch.empty(1, 0, 2, D, device=DEV, dtype=BF),
                          torch.empty(1, 0, 2, D, device=DEV, dtype=BF))
    assert o.shape == (1, 0, 8, D)
