"""-m gpu: the single-token decode path (csrc/decode.hip, kernels/decode.py, models/decode.py) -- NF4 / 16-bit GEMV with
LoRA and bias against the exact fp32 product (and never further from it than the rounding points of bitsandbytes'
naive 4-bit GEMV the reference calls), RoPE + cache append against the training RoPE kernel, split-KV attention against
an fp32 softmax over the cache, and the engine's logits against the training-path forward of the same model."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def _nf4(N, K, seed, dtype, nested=True):
    from unsloth_amd.nf4 import quantize_nf4
    W = (torch.randn(N, K, generator=g(seed)) * 0.05).to(dtype)
    packed, qs = quantize_nf4(W.to(DEV), compress_statistics=nested)
    qs.dtype = dtype
    return packed, qs, R.nf4_fp32_weight(packed, qs)          # exact fp32 value of every code


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nested", [True, False])
@pytest.mark.parametrize("Ns,K,lora", [((256,), 256, False), ((4096, 1024, 1024), 4096, True), ((1024, 1024), 14336, True),
                                       ((40,), 2080, True), ((5, 3, 2, 9), 8192, False)])
def test_gemv_nf4_groups_lora_bias(dtype, nested, Ns, K, lora):
    from unsloth_amd.kernels import decode as D
    x = torch.randn(K, generator=g(1)).to(dtype)
    projs, wants, naive = [], [], []
    for i, N in enumerate(Ns):
        packed, qs, W32 = _nf4(N, K, 10 + i, dtype, nested)
        A = B = s = None
        want = W32.double() @ x.double()
        if lora:
            A = (torch.randn(16, K, generator=g(20 + i)) * 0.05)
            B = (torch.randn(N, 16, generator=g(30 + i)) * 0.05)
            s = 2.0
            want = want + s * (B.double() @ (A.to(dtype).double() @ x.double()))
        bias = (torch.randn(N, generator=g(40 + i)) * 0.1).to(dtype) if i == 0 else None
        if bias is not None:
            want = want + bias.double()
        projs.append((packed, qs, None if A is None else torch.nn.Parameter(A.to(DEV)),
                      None if B is None else torch.nn.Parameter(B.to(DEV)), s, None if bias is None else bias.to(DEV)))
        wants.append(want)
        # the reference kernel's rounding points on the base product (rows of whole 64-code blocks only)
        if K % 64 == 0:
            am = W32.abs().view(N, K // 64, 64).max(dim=2).values
            code = W32 / am.repeat_interleave(64, dim=1).clamp_min(1e-30)
            naive.append(R.gemv_4bit_naive(x, code, am, dtype).double())
        else:
            naive.append(None)
    ys = D.linear_group(x.to(DEV), projs)
    ys2 = D.linear_group(x.to(DEV), projs)
    for y, y2, want, nv, N, p in zip(ys, ys2, wants, naive, Ns, projs):
        assert y.shape == (N,) and torch.equal(y, y2)                       # run-to-run deterministic
        scale = want.abs().max().item() + 1e-6
        err = (y.double().cpu() - want).abs().max().item() / scale
        assert err < (6e-3 if dtype == torch.bfloat16 else 1.5e-3), err    # output rounding + 16-bit code table
        if p[2] is None and p[5] is None and nv is not None:
            base_want = want
            err_naive = (nv - base_want).abs().max().item() / scale
            assert err <= 1.25 * err_naive + 2e-3, (err, err_naive)        # never further from the truth than bnb's kernel


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(128, 512), (1000, 4096), (3, 14336), (128256 // 8, 4096)])
def test_gemv_dense_and_fp32_out(dtype, N, K):
    from unsloth_amd.kernels import decode as D
    W = (torch.randn(N, K, generator=g(3)) * 0.05).to(dtype)
    x = torch.randn(K, generator=g(4)).to(dtype)
    (y,) = D.gemv(x.to(DEV), [dict(W=W.to(DEV), N=N, y_f32=True)], nf4=False)
    want = W.double() @ x.double()
    assert y.dtype == torch.float32
    torch.testing.assert_close(y.double().cpu(), want, rtol=1e-4, atol=1e-4 * want.abs().max().item())
    (y16,) = D.gemv(x.to(DEV), [dict(W=W.to(DEV), N=N)], nf4=False)
    assert torch.equal(y16.cpu(), y.cpu().to(dtype))                      # same sum, one rounding


def test_fast_linear_forward_and_fast_gemv_entry_points():
    """utils.py:872-977 / :1082-1125 call shapes: X [1, 1, K] -> [1, 1, N]; bsz > 1 or q_len > 1 take matmul_lora."""
    from unsloth_amd.kernels import fast_gemv, fast_linear_forward
    from unsloth_amd.nf4 import Linear4bit
    dtype = torch.bfloat16
    lin = torch.nn.Linear(256, 384, bias=True).to(dtype)
    q = Linear4bit.from_linear(lin.to(DEV), 64, True)
    X = torch.randn(1, 1, 256, generator=g(5)).to(dtype).to(DEV)
    W32 = R.nf4_fp32_weight(q.weight.data, q.weight.quant_state)
    want = W32.double() @ X.view(-1).double().cpu()
    y = fast_gemv(X, q.weight, q.weight.quant_state)
    assert y.shape == (1, 1, 384)
    torch.testing.assert_close(y.view(-1).double().cpu(), want, rtol=2e-2, atol=2e-2 * want.abs().max().item())
    y2 = fast_linear_forward(q, X)                                         # plain 4-bit layer: + bias
    torch.testing.assert_close(y2.view(-1).double().cpu(), want + lin.bias.double().cpu(), rtol=2e-2,
                               atol=2e-2 * want.abs().max().item())
    X3 = torch.randn(2, 3, 256, generator=g(6)).to(dtype).to(DEV)
    y3 = fast_linear_forward(q, X3)
    assert y3.shape == (2, 3, 384)
    want3 = X3.double().cpu() @ W32.double().t() + lin.bias.double().cpu()
    torch.testing.assert_close(y3.double().cpu(), want3, rtol=3e-2, atol=3e-2 * want3.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,Hk", [(1, 8, 2), (3, 4, 4)])
def test_rope_kv_append_matches_training_rope(dtype, B, Hq, Hk):
    from unsloth_amd.kernels import decode as Dk
    from unsloth_amd.kernels.rope_embedding import fast_rope_embedding
    D, S = 128, 256
    qkv = torch.randn(B, (Hq + 2 * Hk) * D, generator=g(7)).to(dtype).to(DEV)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(S).float()[:, None] * inv[None, :]
    cos = torch.cat([ang.cos(), ang.cos()], dim=1).to(dtype).to(DEV)
    sin = torch.cat([ang.sin(), ang.sin()], dim=1).to(dtype).to(DEV)
    kv_len = torch.tensor([5, 17, 200][:B], dtype=torch.int32, device=DEV)
    kc = torch.zeros(B, Hk, S, D, dtype=dtype, device=DEV)
    vc = torch.zeros_like(kc)
    ref = qkv.clone()
    Qr = ref[:, :Hq * D].view(B, 1, Hq, D).transpose(1, 2)
    Kr = ref[:, Hq * D:(Hq + Hk) * D].view(B, 1, Hk, D).transpose(1, 2)
    fast_rope_embedding(Qr, Kr, cos, sin, kv_len.clone())                  # in place on `ref`, positions = kv_len
    Dk.rope_kv_append(qkv, cos, sin, kv_len, kc, vc, Hq, Hk, D)
    assert torch.equal(qkv, ref)                                           # same arithmetic, bit for bit
    for b in range(B):
        L = int(kv_len[b])
        assert torch.equal(kc[b, :, L], qkv[b, Hq * D:(Hq + Hk) * D].view(Hk, D))
        assert torch.equal(vc[b, :, L], qkv[b, (Hq + Hk) * D:].view(Hk, D))
        assert float(kc[b, :, :L].abs().sum()) == 0 and float(kc[b, :, L + 1:].abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hk", [(4, 4), (4, 2), (8, 2), (8, 1), (7, 1), (28, 4), (6, 2), (5, 1)])
@pytest.mark.parametrize("lens,window", [((1,), 0), ((16, 129), 0), ((1000, 37, 512), 0), ((700,), 256), ((100,), 256)])
def test_attn_decode_matches_fp32_softmax(dtype, Hq, Hk, lens, window):
    from unsloth_amd.kernels import decode as Dk
    D, S, B = 128, 1024, len(lens)
    G = Hq // Hk
    q = torch.randn(B, Hq * D, generator=g(8)).to(dtype)
    kc = torch.randn(B, Hk, S, D, generator=g(9)).to(dtype)
    vc = torch.randn(B, Hk, S, D, generator=g(10)).to(dtype)
    kv_len = torch.tensor([l - 1 for l in lens], dtype=torch.int32)        # len_add = 1: the new token is already appended
    part = torch.empty(B, Hq, S // 128, D + 2, dtype=torch.float32, device=DEV)
    out = torch.empty(B, Hq * D, dtype=dtype, device=DEV)
    Dk.attn_decode(q.to(DEV), kc.to(DEV), vc.to(DEV), kv_len.to(DEV), out, part, 128, 1.0 / math.sqrt(D), len_add=1,
                   window=window)
    for b, L in enumerate(lens):
        first = L - window if (window and L > window) else 0
        for h in range(Hq):
            k = kc[b, h // G, first:L].double()
            v = vc[b, h // G, first:L].double()
            s = (k @ q[b, h * D:(h + 1) * D].double()) / math.sqrt(D)
            want = torch.softmax(s, dim=0) @ v
            got = out[b, h * D:(h + 1) * D].double().cpu()
            assert (got - want).abs().max().item() < (1.2e-2 if dtype == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hk", [(4, 4), (8, 2), (32, 8), (7, 1), (28, 4)])
@pytest.mark.parametrize("lens,window,S", [((0,), 0, 512), ((1, 127, 128), 0, 512), ((255, 300, 511), 0, 512), ((129, 400), 96, 512),
                                           ((20,), 96, 512), ((1500, 100, 2046), 0, 2048)])
def test_attn_decode_fused_is_the_three_launches(dtype, Hq, Hk, lens, window, S):
    """uamd_attn_decode_fused (RoPE + append + split attention + combine in ONE launch) against uamd_rope_kv_append ->
    uamd_attn_decode: the same cache bit for bit (same RoPE arithmetic), the same output to the rounding of the output dtype (the
    keys of a split are accumulated chunk-wise instead of one by one), over three consecutive tokens (tags / arrival counters
    take care of themselves), with the raw q|k|v row left untouched. lens = tokens already in the cache (0: the first one;
    127 / 128 / 255: the new key is the last of a split / the first of the next / the cache's last slot). Launches of up to 256
    workgroups combine through {value, tag} granules, larger ones (the S = 2048 case with 8 KV heads: 16 x 8 x 3 = 384) through the
    arrival counter: both paths are in the grid."""
    from unsloth_amd.kernels import decode as Dk
    D, B = 128, len(lens)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(S).float()[:, None] * inv[None, :]
    cos = torch.cat([ang.cos(), ang.cos()], dim=1).to(dtype).to(DEV)
    sin = torch.cat([ang.sin(), ang.sin()], dim=1).to(dtype).to(DEV)
    kc1 = torch.randn(B, Hk, S, D, generator=g(9)).to(dtype).to(DEV)
    vc1 = torch.randn(B, Hk, S, D, generator=g(10)).to(dtype).to(DEV)
    kc2, vc2 = kc1.clone(), vc1.clone()
    kv_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    part1 = torch.empty(B, Hq, S // 128, D + 2, dtype=torch.float32, device=DEV)
    part2, cnt = Dk.fused_attn_workspace(B, Hq, Hk, S, D, 128, DEV)
    scale = 1.0 / math.sqrt(D)
    for step in range(3):
        if max(lens) + step >= S:                 # the cache is full (lens 511 of 512: one token into its last slot, then stop)
            break
        raw = torch.randn(B, (Hq + 2 * Hk) * D, generator=g(20 + step)).to(dtype).to(DEV)
        q1 = raw.clone()
        out1 = torch.empty(B, Hq * D, dtype=dtype, device=DEV)
        Dk.rope_kv_append(q1, cos, sin, kv_len, kc1, vc1, Hq, Hk, D)
        Dk.attn_decode(q1[:, :Hq * D], kc1, vc1, kv_len, out1, part1, 128, scale, len_add=1, window=window)
        keep = raw.clone()
        out2 = torch.full((B, Hq * D), float("nan"), dtype=dtype, device=DEV)
        Dk.attn_decode_fused(raw, cos, sin, kv_len, kc2, vc2, out2, part2, cnt, 128, scale, Hq, window=window)
        assert torch.equal(raw, keep)
        err = (out2.float() - out1.float()).abs().max().item()
        assert err <= (1.6e-2 if dtype == torch.bfloat16 else 2e-3) * max(out1.float().abs().max().item(), 1e-3), (step, err)
        assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
        assert int(cnt.abs().sum()) == 0                      # arrival counters (large launches) back at zero
        kv_len += 1


@pytest.mark.parametrize("split_keys,lens", [(256, (700, 255, 256)), (512, (1000,)), (64, (130, 64))])
def test_attn_decode_fused_with_other_split_sizes(split_keys, lens):
    """The engine gives long contexts longer splits (<= 256 workgroups per launch): several 128-key chunks per workgroup, and the
    64-key case where a chunk is half empty."""
    from unsloth_amd.kernels import decode as Dk
    dtype, D, S, Hq, Hk, B = torch.bfloat16, 128, 1024, 8, 2, len(lens)
    cos = torch.randn(S, D, generator=g(1)).clamp(-1, 1).to(dtype).to(DEV)
    sin = torch.randn(S, D, generator=g(2)).clamp(-1, 1).to(dtype).to(DEV)
    kc1 = torch.randn(B, Hk, S, D, generator=g(9)).to(dtype).to(DEV)
    vc1 = torch.randn(B, Hk, S, D, generator=g(10)).to(dtype).to(DEV)
    kc2, vc2 = kc1.clone(), vc1.clone()
    kv_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    part1 = torch.empty(B, Hq, S // split_keys, D + 2, dtype=torch.float32, device=DEV)
    part2, cnt = Dk.fused_attn_workspace(B, Hq, Hk, S, D, split_keys, DEV)
    raw = torch.randn(B, (Hq + 2 * Hk) * D, generator=g(3)).to(dtype).to(DEV)
    q1 = raw.clone()
    out1 = torch.empty(B, Hq * D, dtype=dtype, device=DEV)
    out2 = torch.full_like(out1, float("nan"))
    Dk.rope_kv_append(q1, cos, sin, kv_len, kc1, vc1, Hq, Hk, D)
    Dk.attn_decode(q1[:, :Hq * D], kc1, vc1, kv_len, out1, part1, split_keys, 1.0 / math.sqrt(D), len_add=1)
    Dk.attn_decode_fused(raw, cos, sin, kv_len, kc2, vc2, out2, part2, cnt, split_keys, 1.0 / math.sqrt(D), Hq)
    assert torch.equal(kc2, kc1) and torch.equal(vc2, vc1)
    assert (out2.float() - out1.float()).abs().max().item() <= 1.6e-2 * out1.float().abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nf4", [True, False])
def test_gemv_fused_glu_epilogue_and_the_in_launch_lora_hand_off(dtype, nf4):
    """uamd_gemv_fused with glu: gate | up in, h = SwiGLU out -- bit-identical to the gate|up launch + uamd_swiglu_fg when there
    is no adapter (same dot products, same rounding points), within the summation-order noise of t = A x with one. And the
    hand-off workspace: 40 launches over alternating tokens and shapes sharing it reproduce their first results bit for bit (a
    stale or torn {value, tag} granule, or a tag that is used twice, shows up as a different or NaN output)."""
    from unsloth_amd.kernels import decode as D
    from unsloth_amd.kernels.swiglu import swiglu_fg_kernel
    K, N, r = 1024, 1408, 8
    gen = g(91)

    def weight(n, seed):
        if nf4:
            W, qs, _ = _nf4(n, K, seed, dtype)
            return W, qs
        return (torch.randn(n, K, generator=g(seed)) * 0.05).to(dtype).to(DEV), None
    bare, lora = [], []
    for i in range(2):
        W, qs = weight(N, 300 + i)
        bias = (torch.randn(N, generator=gen) * 0.1).to(dtype).to(DEV) if i == 0 else None
        bare.append((W, qs, None, None, None, bias))
        lora.append((W, qs, (torch.randn(r, K, generator=gen) * 0.05).to(DEV), (torch.randn(N, r, generator=gen) * 0.05).to(DEV),
                     2.0, bias))
    xs = [torch.randn(K, generator=gen).to(dtype).to(DEV) for _ in range(2)]
    for x in xs:
        e, gg = D.linear_group(x, bare)
        want = swiglu_fg_kernel(e.view(1, 1, N), gg.view(1, 1, N)).view(-1)
        (h,) = D.linear_group(x, bare, fused=dict(mode=0, glu=True))
        assert h.shape == (N,) and torch.equal(h, want)
        e, gg = D.linear_group(x, lora)
        want = swiglu_fg_kernel(e.view(1, 1, N), gg.view(1, 1, N)).view(-1)
        (h,) = D.linear_group(x, lora, fused=dict(mode=0, glu=True))
        assert (h.float() - want.float()).abs().max() <= 3e-2 * want.float().abs().max()
    W3, qs3 = weight(384, 310)
    other = [(W3, qs3, (torch.randn(16, K, generator=gen) * 0.05).to(DEV), (torch.randn(384, 16, generator=gen) * 0.05).to(DEV), 0.5, None)]
    first = {}
    for it in range(40):
        x = xs[it & 1]
        which = (it // 2) % 3
        if which == 0:
            (y,) = D.linear_group(x, lora, fused=dict(mode=0, glu=True))
        elif which == 1:
            y = torch.cat(D.linear_group(x, lora, fused=dict(mode=0)))
        else:
            (y,) = D.linear_group(x, other, fused=dict(mode=0))
        assert bool(torch.isfinite(y.float()).all())
        key = (it & 1, which)
        if key in first:
            assert torch.equal(y, first[key]), (it, key)
        else:
            first[key] = y.clone()
    assert D.sync_workspace(x.device)._host >= 40          # one host tag per launch with adapters


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nf4", [True, False])
@pytest.mark.parametrize("K,Ns,r", [(14336, (512,), 16), (6144, (96, 40), 8), (4096, (1024, 256, 256), 16), (2080, (40,), 64)])
def test_gemv_fused_lora_hand_off_at_long_rows(dtype, nf4, K, Ns, r):
    """uamd_gemv_fused, t = A x inside the launch, where a row of A is split over 1 / 2 / 4 waves of a t workgroup (K <= 4096 /
    8192 / 16384) and where a launch makes several trips per wave: against the exact fp64 product, and equal (to the fp32
    summation order of t) to the two-launch path."""
    from unsloth_amd.kernels import decode as D
    gen = g(5)
    x = torch.randn(K, generator=gen).to(dtype)
    projs, wants = [], []
    for i, N in enumerate(Ns):
        if nf4:
            W, qs, W32 = _nf4(N, K, 400 + i, dtype)
        else:
            Wd = (torch.randn(N, K, generator=g(400 + i)) * 0.05).to(dtype)
            W, qs, W32 = Wd.to(DEV), None, Wd.float()
        A = (torch.randn(r, K, generator=gen) * 0.05)
        B = (torch.randn(N, r, generator=gen) * 0.05)
        projs.append((W, qs, torch.nn.Parameter(A.to(DEV)), torch.nn.Parameter(B.to(DEV)), 2.0, None))
        wants.append(W32.double().cpu() @ x.double() + 2.0 * (B.double() @ (A.to(dtype).double() @ x.double())))
    got = D.linear_group(x.to(DEV), projs, fused=dict(mode=0))
    two = D.linear_group(x.to(DEV), projs)
    for y, y2, want, N in zip(got, two, wants, Ns):
        assert y.shape == (N,)
        scale = want.abs().max().item() + 1e-6
        assert (y.double().cpu() - want).abs().max().item() / scale < (6e-3 if dtype == torch.bfloat16 else 1.5e-3)
        assert (y.float() - y2.float()).abs().max().item() <= 1e-2 * scale


@pytest.mark.parametrize("rows,n", [(1, 128256), (3, 1000), (2, 64), (1, 1), (4, 32000)])
def test_argmax_kernel_is_torch_argmax_with_first_index_ties(rows, n):
    from unsloth_amd.kernels import decode as D
    x = torch.randn(rows, n, generator=g(3)).to(DEV)
    assert torch.equal(D.argmax_f32(x), torch.argmax(x, dim=-1))
    x[:, n // 3:] = x[:, n // 3:].clamp(max=0.5)
    x[:, n // 3] = 7.0
    if n > 5:
        x[:, n - 2] = 7.0                          # a tie: the first maximal element wins
    assert torch.equal(D.argmax_f32(x), torch.full((rows,), n // 3, dtype=torch.long, device=DEV))
    out = torch.empty(rows, dtype=torch.long, device=DEV)
    ws = (torch.empty(rows * 64, device=DEV), torch.empty(rows * 64, dtype=torch.long, device=DEV))
    assert D.argmax_f32(x, out=out, ws=ws) is out and int(out[0]) == n // 3


def _tiny(load_in_4bit=True, r=8):
    from transformers import LlamaConfig
    from unsloth_amd import FastLanguageModel
    cfg = LlamaConfig(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=128, vocab_size=1000, rms_norm_eps=1e-5, max_position_embeddings=512,
                      rope_parameters={"rope_type": "default", "rope_theta": 5e5}, tie_word_embeddings=False)
    model, _ = FastLanguageModel.from_pretrained(config=cfg, max_seq_length=256, load_in_4bit=load_in_4bit, device=DEV,
                                                 random_state=3407, use_gradient_checkpointing=False)
    model = FastLanguageModel.get_peft_model(model, r=r, lora_alpha=2 * r, use_gradient_checkpointing=False, random_state=3407)
    gg = torch.Generator().manual_seed(11)
    for n, p in model.named_parameters():
        if "lora_B" in n:
            p.data.copy_((torch.randn(p.shape, generator=gg) * 0.05).to(DEV))
    return model


@pytest.mark.parametrize("load_in_4bit", [True, False])
def test_engine_logits_match_training_path_forward(load_in_4bit, monkeypatch):
    """Prefill + 6 graph-replayed steps: every step's logits equal (bf16 noise) the last-position logits of the
    training-path forward over the same prefix; eager and hipGraph steps agree bit for bit; greedy generate is the
    argmax chain."""
    from unsloth_amd.models.decode import DecodeEngine
    monkeypatch.setenv("UNSLOTH_RETURN_LOGITS", "1")
    model = _tiny(load_in_4bit)
    model.eval()
    ids = torch.randint(0, 1000, (1, 21), generator=g(12)).to(DEV)
    eng = DecodeEngine(model, max_seq_len=256, batch=1, use_graph=True)
    eng_e = DecodeEngine(model, max_seq_len=256, batch=1, use_graph=False)
    lg, lg_e = eng.prefill(ids), eng_e.prefill(ids)
    assert torch.equal(lg, lg_e)
    seq = ids
    for step in range(6):
        with torch.no_grad():
            full = model(input_ids=seq).logits[:, -1].float()
        scale = full.abs().max().item()
        assert (lg - full).abs().max().item() < 4e-2 * scale, (step, (lg - full).abs().max().item(), scale)
        nxt = torch.argmax(lg, dim=-1)
        seq = torch.cat([seq, nxt.view(1, 1)], dim=1)
        lg, lg_e = eng.step(nxt).clone(), eng_e.step(nxt).clone()
        assert torch.equal(lg, lg_e), f"graph replay differs from the eager step at step {step}"
    assert int(eng.kv_len[0]) == 27
    # a second prompt on the SAME engine (graph already captured): the replayed steps must return the graph's logits,
    # not the tensor prefill() produced
    lg2 = eng.prefill(ids)
    assert bool(torch.isfinite(lg2).all()) and lg2.device == lg_e.device
    first = torch.argmax(lg2, dim=-1)
    again = eng.step(first).clone()
    eng_f = DecodeEngine(model, max_seq_len=256, batch=1, use_graph=False)
    eng_f.prefill(ids)
    assert torch.equal(again, eng_f.step(first))
    out = DecodeEngine(model, max_seq_len=256).generate(ids, max_new_tokens=6)
    assert torch.equal(out[:, :27], seq[:, :27])


def test_for_inference_generate_and_back_to_training():
    """FastLanguageModel.for_inference(model): model.generate is the decode engine (greedy = the engine's argmax chain,
    sampling reproducible under a seeded generator); for_training restores HF's generate and training still works."""
    from unsloth_amd import FastLanguageModel
    from unsloth_amd.models.decode import DecodeEngine
    model = _tiny(True)
    FastLanguageModel.for_inference(model)
    assert not model.training and hasattr(model, "_old_generate")
    ids = torch.randint(0, 1000, (1, 9), generator=g(13)).to(DEV)
    out = model.generate(input_ids=ids, max_new_tokens=5)
    want = DecodeEngine(model, max_seq_len=128).generate(ids, max_new_tokens=5)
    assert out.shape == (1, 14) and torch.equal(out, want)
    eng = DecodeEngine(model, max_seq_len=128)
    s1 = eng.generate(ids, max_new_tokens=6, do_sample=True, temperature=0.8, top_k=20, generator=torch.Generator(DEV).manual_seed(3))
    s2 = eng.generate(ids, max_new_tokens=6, do_sample=True, temperature=0.8, top_k=20, generator=torch.Generator(DEV).manual_seed(3))
    assert torch.equal(s1, s2) and s1.shape == (1, 15)
    # early stop on an eos id: the first generated token of the greedy chain
    eos = int(out[0, 9])
    short = eng.generate(ids, max_new_tokens=5, eos_token_id=eos)
    assert short.shape == (1, 10)
    FastLanguageModel.for_training(model, use_gradient_checkpointing=False)
    assert model.training and not hasattr(model, "_old_generate")
    lab = ids.clone()
    loss = model(input_ids=ids, labels=lab).loss
    loss.backward()
    assert torch.isfinite(loss)


def test_generate_train_generate_uses_the_trained_factors():
    """the decode path caches activation-dtype copies of the LoRA A factors; FlatAdamW updates the
    parameters with a raw HIP kernel that never bumps Parameter._version. generate -> a few optimizer steps -> generate
    must decode with the TRAINED factors: the engine's logits equal the training-path forward of the updated model, both
    with a fresh engine and with the engine (and its captured hipGraph) that was alive during training."""
    from unsloth_amd import FastLanguageModel
    from unsloth_amd.models.decode import DecodeEngine
    from unsloth_amd.trainer import make_optimizer, training_step
    model = _tiny(True)
    ids = torch.randint(0, 1000, (1, 17), generator=g(21)).to(DEV)
    FastLanguageModel.for_inference(model)
    before = model.generate(input_ids=ids, max_new_tokens=4)
    eng_live = DecodeEngine(model, max_seq_len=128)                      # stays alive across the training steps
    lg0 = eng_live.prefill(ids).clone()
    eng_live.step(torch.argmax(lg0, dim=-1))                             # captures the graph
    FastLanguageModel.for_training(model, use_gradient_checkpointing=False)
    opt = make_optimizer(model, lr=5e-2)                                 # FlatAdamW: large steps so the logits move
    assert type(opt).__name__ == "FlatAdamW"
    tr = torch.randint(0, 1000, (2, 48), generator=g(22)).to(DEV)
    for _ in range(3):
        training_step(model, dict(input_ids=tr, labels=tr.clone()), opt)
    os.environ["UNSLOTH_RETURN_LOGITS"] = "1"
    try:
        model.eval()
        with torch.no_grad():
            full = model(input_ids=ids).logits[:, -1].float()
    finally:
        os.environ.pop("UNSLOTH_RETURN_LOGITS")
    scale = full.abs().max().item()
    assert (full - lg0.float()).abs().max().item() > 0.1 * scale, "training did not move the logits: test is vacuous"
    for eng in (DecodeEngine(model, max_seq_len=128), eng_live):
        lg = eng.prefill(ids).float()
        assert (lg - full).abs().max().item() < 4e-2 * scale
        nxt = torch.argmax(lg, dim=-1)
        step_lg = eng.step(nxt).float().clone()
        with torch.no_grad():
            os.environ["UNSLOTH_RETURN_LOGITS"] = "1"
            try:
                full2 = model(input_ids=torch.cat([ids, nxt.view(1, 1)], dim=1)).logits[:, -1].float()
            finally:
                os.environ.pop("UNSLOTH_RETURN_LOGITS")
        assert (step_lg - full2).abs().max().item() < 4e-2 * full2.abs().max().item(), "decode step used stale LoRA factors"
    FastLanguageModel.for_inference(model)
    after = model.generate(input_ids=ids, max_new_tokens=4)
    assert after.shape == before.shape


def test_generate_raises_on_poisoned_logits_instead_of_emitting_tokens():
    """a timed-out in-launch hand-off writes NaN on purpose; generate() must turn that into an error, not into token
    ids. The poison is injected where a timeout would put it: the engine's step output."""
    from unsloth_amd import FastLanguageModel
    from unsloth_amd.models.decode import DecodeEngine
    model = _tiny(True)
    FastLanguageModel.for_inference(model)
    ids = torch.randint(0, 1000, (1, 7), generator=g(31)).to(DEV)
    eng = DecodeEngine(model, max_seq_len=64, batch=1, use_graph=False)
    good = eng.generate(ids, max_new_tokens=4)
    assert good.shape == (1, 11)
    real = eng.step
    calls = []

    def poisoned(tok):
        lg = real(tok)
        calls.append(1)
        return lg * float("nan") if len(calls) == 2 else lg
    eng.step = poisoned
    with pytest.raises(RuntimeError, match="non-finite logits"):
        eng.generate(ids, max_new_tokens=5)


def test_fast_generate_kwargs_follow_hf_semantics():
    """eos / pad default from generation_config, finished rows emit the pad id, max_length is
    honoured, and arguments the engine does not implement (repetition_penalty, a padded attention_mask, ...) go to HF's
    generate instead of being dropped. Nucleus sampling (top_p, also when only the checkpoint's generation_config asks for
    it, as Llama-3-Instruct's does) stays on the engine since round 5."""
    from unsloth_amd import FastLanguageModel
    model = _tiny(True)
    FastLanguageModel.for_inference(model)
    ids = torch.randint(0, 1000, (2, 9), generator=g(23)).to(DEV)
    free = model.generate(input_ids=ids, max_new_tokens=6)
    eos = int(free[0, 9])                                   # row 0 stops after its first token
    base = model.get_base_model()
    base.generation_config.eos_token_id = eos
    base.generation_config.pad_token_id = 7
    model.generation_config = base.generation_config
    out = model.generate(input_ids=ids, max_new_tokens=6)
    assert out[0, 9] == eos and bool((out[0, 10:] == 7).all()), out[0]
    assert model.generate(input_ids=ids, max_length=12).shape[1] <= 12
    calls = []
    model._old_generate = lambda *a, **k: (calls.append(k), free)[1]
    model.generate(input_ids=ids, max_new_tokens=3, repetition_penalty=1.3)
    assert calls and calls[-1].get("repetition_penalty") == 1.3
    mask = torch.ones_like(ids)
    mask[1, :3] = 0
    model.generate(input_ids=ids, max_new_tokens=3, attention_mask=mask)
    assert len(calls) == 2 and calls[-1].get("attention_mask") is mask
    model.generate(input_ids=ids, max_new_tokens=3, attention_mask=torch.ones_like(ids), top_p=1.0)
    assert len(calls) == 2                                   # neutral values stay on the engine
    # nucleus sampling on the engine: a tiny top_p keeps only the most likely token -> the greedy continuation
    base.generation_config.eos_token_id = None
    greedy = model.generate(input_ids=ids, max_new_tokens=4)
    sampled = model.generate(input_ids=ids, max_new_tokens=4, do_sample=True, top_p=1e-4, generator=torch.Generator(device=DEV).manual_seed(3))
    assert len(calls) == 2 and torch.equal(greedy, sampled)
    base.generation_config.do_sample, base.generation_config.top_p, base.generation_config.temperature = True, 0.9, 0.6
    out = model.generate(input_ids=ids, max_new_tokens=4)   # the checkpoint's own sampling settings: still the engine
    assert len(calls) == 2 and out.shape == greedy.shape


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("wdtype", ["same", "fp32"])
@pytest.mark.parametrize("K", [1024, 8192])          # 8192: rows longer than a thread's register vector (the two-pass prologue)
def test_gemv_fused_prologues_match_the_separate_launches(dtype, wdtype, K):
    """uamd_gemv_fused: the token produced inside the launch (SwiGLU / residual add + RMSNorm, bit-identical x to the
    separate kernels) and the LoRA t = A x computed by the launch's own first workgroups."""
    from unsloth_amd.kernels import decode as D
    from unsloth_amd.kernels.rms_layernorm import add_rms_fwd, rms_fwd
    from unsloth_amd.kernels.swiglu import swiglu_fg_kernel
    Ns, r = (512, 256, 256), 8
    gen = g(77)
    projs = []
    for i, N in enumerate(Ns):
        W, qs, _ = _nf4(N, K, 100 + i, dtype)
        A = (torch.randn(r, K, generator=gen) * 0.05).to(DEV)
        B = (torch.randn(N, r, generator=gen) * 0.05).to(DEV)
        projs.append((W, qs, A, B, 2.0, None))
    a = (torch.randn(K, generator=gen)).to(dtype).to(DEV)
    res = (torch.randn(K, generator=gen)).to(dtype).to(DEV)
    w = (1 + 0.1 * torch.randn(K, generator=gen)).to(torch.float32 if wdtype == "fp32" else dtype).to(DEV)
    # mode 2: residual add + norm
    if K <= 4096:
        h_ref, x_ref, _ = add_rms_fwd(a.view(1, K), res.view(1, K), w, 1e-5)
    else:                                            # (the fused add + norm kernel keeps a row in registers: two ops for long rows)
        h_ref = (a.float() + res.float()).to(dtype).view(1, K)
        x_ref = rms_fwd(h_ref, w, 1e-5)[0]
    want = D.linear_group(x_ref.view(-1), projs)
    h_out = torch.empty_like(res)
    got = D.linear_group(a, projs, fused=dict(mode=2, res=res, norm_w=w, eps=1e-5, h_out=h_out))
    assert torch.equal(h_out, h_ref.view(-1))
    for y, yr in zip(got, want):
        assert (y.float() - yr.float()).abs().max() <= 2e-2 * yr.float().abs().max()
    # mode 2 without a delta (first layer): x = rmsnorm(res)
    x0 = rms_fwd(res.view(1, K), w, 1e-5)[0]
    want0 = D.linear_group(x0.view(-1), projs)
    got0 = D.linear_group(None, projs, fused=dict(mode=2, res=res, norm_w=w, eps=1e-5, h_out=None))
    for y, yr in zip(got0, want0):
        assert (y.float() - yr.float()).abs().max() <= 2e-2 * yr.float().abs().max()
    # mode 1: SwiGLU of two vectors
    hh = swiglu_fg_kernel(a.view(1, 1, K), res.view(1, 1, K)).view(-1)
    want1 = D.linear_group(hh, projs[:1])
    got1 = D.linear_group(a, projs[:1], fused=dict(mode=1, x2=res))
    assert (got1[0].float() - want1[0].float()).abs().max() <= 2e-2 * want1[0].float().abs().max()
    # mode 0 with in-launch A x, and without any adapter
    want2 = D.linear_group(a, projs)
    got2 = D.linear_group(a, projs, fused=dict(mode=0))
    for y, yr in zip(got2, want2):
        assert (y.float() - yr.float()).abs().max() <= 2e-2 * yr.float().abs().max()
    bare = [(p[0], p[1], None, None, None, None) for p in projs]
    for y, yr in zip(D.linear_group(a, bare, fused=dict(mode=0)), D.linear_group(a, bare)):
        assert torch.equal(y, yr)


@pytest.mark.parametrize("load_in_4bit", [True, False])
def test_fused_decode_step_matches_the_separate_launches(load_in_4bit):
    """DecodeEngine with 5 launches per layer (uamd_gemv_fused, uamd_attn_decode_fused) against the 14-launch step: same tokens,
    logits within the fp32 summation-order noise of t = A x."""
    from unsloth_amd.models import decode as MD
    model = _tiny(load_in_4bit)
    model.eval()
    ids = torch.randint(0, 1000, (1, 19), generator=g(5)).to(DEV)
    outs = {}
    for fused in (True, False):
        MD.FUSED_STEP = fused
        try:
            eng = MD.DecodeEngine(model, max_seq_len=128, batch=1, use_graph=fused)
            lg = eng.prefill(ids)
            seq = []
            for _ in range(8):
                nxt = torch.argmax(lg, dim=-1)
                lg = eng.step(nxt).clone()
                seq.append(lg)
            outs[fused] = torch.stack(seq)
        finally:
            MD.FUSED_STEP = False
    scale = outs[False].abs().max().item()
    assert (outs[True] - outs[False]).abs().max().item() <= 2e-2 * scale
    assert torch.equal(outs[True].argmax(-1), outs[False].argmax(-1))
