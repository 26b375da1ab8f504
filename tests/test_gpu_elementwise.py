"""-m gpu: the HIP kernels (through the C ABI, via unsloth_amd.kernels) against the CPU oracle and
against the committed reference-Triton golden fixtures. RMSNorm / RoPE / SwiGLU / GeGLU / CE."""
import pytest
import torch

from oracle import ref_ops as R
from tests._util import assert_ulp

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.bfloat16, torch.float16, torch.float32]


def g(seed):
    return torch.Generator().manual_seed(seed)


def U(dtype, ulps):
    """fp32 comparisons allow 32 ulp (4e-6 relative): rsqrt/exp/erf/tanh implementations differ by a
    few fp32 ulps between the GPU and the CPU oracle; 16-bit results must match to `ulps`."""
    return 32 if dtype == torch.float32 else ulps


# ---------------------------------------------------------------- RMSNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("dim", [512, 1024, 2048, 4096, 8192, 100, 16384])
@pytest.mark.parametrize("gemma", [False, True])
def test_rms_layernorm(dtype, dim, gemma):
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    rows = 37
    X = torch.randn(rows, dim, generator=g(3407)).to(dtype)
    W = torch.rand(dim, generator=g(42)).to(dtype)            # U(0,1) like rms_layernorm.py:314
    dY = torch.randn(rows, dim, generator=g(7)).to(dtype)
    Yo, r = R.rms_layernorm_forward(X, W, 1e-5, gemma)
    dXo = R.rms_layernorm_backward(dY, X, W, r, gemma)
    Xg = X.to(DEV).requires_grad_(True)
    Y = Fast_RMS_Layernorm.apply(Xg, W.to(DEV), 1e-5, gemma)
    assert_ulp(Y, Yo, dtype, ulps=U(dtype, 1), what="rms fwd", allow_frac=2e-3)
    dYg = dY.to(DEV)
    ptr = dYg.data_ptr()
    Y.backward(dYg)
    assert_ulp(Xg.grad, dXo, dtype, ulps=U(dtype, 2), what="rms bwd", allow_frac=2e-3)
    if not gemma:
        assert Xg.grad.data_ptr() == ptr, "non-gemma backward must write dX in place over dY"


def test_rms_layernorm_fp32_weight_bf16_act():
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    X = torch.randn(9, 4096, generator=g(1)).to(torch.bfloat16)
    W = torch.rand(4096, generator=g(2))
    Yo, _ = R.rms_layernorm_forward(X, W, 1e-6)
    Y = Fast_RMS_Layernorm.apply(X.to(DEV), W.to(DEV), 1e-6, False)
    assert_ulp(Y, Yo, torch.bfloat16, ulps=1, what="rms fp32-W", allow_frac=2e-3)


def test_rms_layernorm_matches_hf_module():
    """HF LlamaRMSNorm has the same rounding point -> valid bf16 oracle."""
    from transformers.models.llama.modeling_llama import LlamaRMSNorm
    from unsloth_amd.kernels.rms_layernorm import fast_rms_layernorm
    m = LlamaRMSNorm(2048, eps=1e-5).to(torch.bfloat16)
    m.weight.data = torch.rand(2048, generator=g(5)).to(torch.bfloat16)
    X = torch.randn(4, 21, 2048, generator=g(6)).to(torch.bfloat16)
    want = m(X)
    got = fast_rms_layernorm(m.to(DEV), X.to(DEV))
    assert_ulp(got, want, torch.bfloat16, ulps=1, what="vs HF LlamaRMSNorm", allow_frac=2e-3)


def test_rms_golden(golden):
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    for dn, dt in (("f32", torch.float32), ("f16", torch.float16)):
        for gemma in (0, 1):
            c = golden[f"rms_{dn}_gemma{gemma}"]
            Xg = c["X"].to(DEV).requires_grad_(True)
            Y = Fast_RMS_Layernorm.apply(Xg, c["W"].to(DEV), c["eps"], bool(gemma))
            assert_ulp(Y, c["Y"], dt, ulps=4 if dt == torch.float32 else 1, what=f"golden rms {dn}", allow_frac=5e-3)
            Y.backward(c["dY"].to(DEV))
            assert_ulp(Xg.grad, c["dX"], dt, ulps=64 if dt == torch.float32 else 2, what=f"golden rms bwd {dn}",
                       allow_frac=5e-3)


# ---------------------------------------------------------------- RoPE
def _tables(T, D, dtype, theta=5e5):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    fr = torch.outer(torch.arange(T, dtype=torch.int64).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


@pytest.mark.parametrize("qd,td", [(torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16),
                                   (torch.bfloat16, torch.float32), (torch.float32, torch.float32)])
@pytest.mark.parametrize("D", [128, 64, 80])
def test_rope_qk_indexed_and_dense(qd, td, D):
    from unsloth_amd.kernels.rope_embedding import fast_rope_embedding
    B, Hq, Hk, T = 2, 8, 2, 50
    cos, sin = _tables(256, D, td)
    Q = torch.randn(B, T, Hq, D, generator=g(3)).to(qd)
    K = torch.randn(B, T, Hk, D, generator=g(4)).to(qd)
    # packed documents: positions restart (int32), as the padding-free collator emits
    idx = torch.cat([torch.arange(30), torch.arange(20), torch.arange(50)]).to(torch.int32)
    Qv, Kv = Q.transpose(1, 2), K.transpose(1, 2)             # strided [B,H,T,D] views
    for indices in (idx, None):
        Qo, Ko = R.rope_embedding_qk(Qv, Kv, cos, sin, indices)
        Qg, Kg = Q.to(DEV).transpose(1, 2), K.to(DEV).transpose(1, 2)
        Qr, Kr = fast_rope_embedding(Qg, Kg, cos.to(DEV), sin.to(DEV), None if indices is None else indices.to(DEV))
        native = qd == td and qd != torch.float32
        assert_ulp(Qr, Qo, qd, ulps=0 if native else 1, atol=0 if native else None, what="rope Q")
        assert_ulp(Kr, Ko, qd, ulps=0 if native else 1, atol=0 if native else None, what="rope K")
        assert Qr.data_ptr() == Qg.data_ptr(), "rotation must be in place on the strided view"


def test_rope_backward_is_inverse_rotation():
    from unsloth_amd.kernels.rope_embedding import Fast_RoPE_Embedding_QK, Fast_RoPE_Embedding
    B, H, Hk, T, D = 1, 32, 8, 2048, 128
    cos, sin = _tables(T, D, torch.float32)
    Q = torch.randn(B, H, T, D, generator=g(9)).to(torch.bfloat16).to(DEV)
    K = torch.randn(B, Hk, T, D, generator=g(10)).to(torch.bfloat16).to(DEV)
    Q0, K0 = Q.clone(), K.clone()
    Qg, Kg = (Q * 1.0).requires_grad_(True), (K * 1.0).requires_grad_(True)
    qo, ko = Fast_RoPE_Embedding_QK.apply(Qg * 1.0, Kg * 1.0, cos.to(DEV), sin.to(DEV), None)
    # gradient of sum(out * out_detached) = rotate^T(out) = original input (rotation is orthogonal)
    torch.autograd.backward([qo, ko], [qo.detach().clone(), ko.detach().clone()])
    from tests._util import rel_fro
    assert rel_fro(Qg.grad, Q0) < 8e-3 and rel_fro(Kg.grad, K0) < 8e-3, (rel_fro(Qg.grad, Q0), rel_fro(Kg.grad, K0))
    # dense entry point agrees with the strided one
    Qd = Fast_RoPE_Embedding.apply(Q0.transpose(1, 2).contiguous(), cos.to(DEV), sin.to(DEV)).transpose(1, 2)
    assert torch.equal(Qd, qo.detach())


def test_rope_golden(golden):
    from unsloth_amd.kernels.rope_embedding import Fast_RoPE_Embedding_QK
    for dn, dt in (("f32", torch.float32), ("f16", torch.float16)):
        c = golden[f"rope_{dn}"]
        cos, sin = c["cos"].to(DEV), c["sin"].to(DEV)
        Qr, Kr = Fast_RoPE_Embedding_QK.apply(c["Q"].to(DEV), c["K"].to(DEV), cos, sin, c["idx"].to(DEV))
        assert_ulp(Qr, c["Q_idx"], dt, ulps=2 if dt == torch.float32 else 0, atol=1e-7 if dt == torch.float32 else 0, what="golden rope Q")
        assert_ulp(Kr, c["K_idx"], dt, ulps=2 if dt == torch.float32 else 0, atol=1e-7 if dt == torch.float32 else 0, what="golden rope K")


# ---------------------------------------------------------------- GLU
KINDS = {"swiglu": ("swiglu_fg_kernel", "swiglu_DWf_DW_dfg_kernel"),
         "geglu_exact": ("geglu_exact_forward_kernel", "geglu_exact_backward_kernel"),
         "geglu_approx": ("geglu_approx_forward_kernel", "geglu_approx_backward_kernel")}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("shape", [(2, 33, 1024), (1, 7, 333), (1, 2048, 14336)])
def test_glu(dtype, kind, shape):
    import unsloth_amd.kernels as K
    fwd, bwd = getattr(K, KINDS[kind][0]), getattr(K, KINDS[kind][1])
    e = torch.randn(*shape, generator=g(11)).to(dtype)
    gg = torch.randn(*shape, generator=g(12)).to(dtype)
    DW = torch.randn(shape[0] * shape[1], shape[2], generator=g(13)).to(dtype)
    h = fwd(e.to(DEV), gg.to(DEV))
    # fp32: 1 + erf(x) / 1 + tanh(x) cancel for x << 0, so the result carries the ABSOLUTE accuracy of the
    # transcendental (about 1 ulp of 1.0, times |e g|), not a relative one: allow 1e-6 absolute on O(1) data.
    at = 1e-6 if dtype == torch.float32 else None
    assert_ulp(h, R.glu_forward(e, gg, kind), dtype, ulps=U(dtype, 1), atol=at, what=f"{kind} fwd", allow_frac=5e-3)
    e2, g2, d2 = e.view(-1, shape[2]).to(DEV), gg.view(-1, shape[2]).to(DEV), DW.to(DEV)
    ptrs = (d2.data_ptr(), e2.data_ptr(), g2.data_ptr())
    ho, dfo, deo = R.glu_backward(DW, e.view(-1, shape[2]), gg.view(-1, shape[2]), kind)
    hh, df, de = bwd(d2, e2, g2)
    assert (hh.data_ptr(), df.data_ptr(), de.data_ptr()) == ptrs, "backward must overwrite DW, e, g"
    at = 4e-6 if dtype == torch.float32 else None      # |DW| multiplies the same absolute error
    assert_ulp(hh, ho, dtype, ulps=U(dtype, 1), atol=at, what=f"{kind} bwd h", allow_frac=5e-3)
    assert_ulp(df, dfo, dtype, ulps=U(dtype, 1), atol=at, what=f"{kind} bwd df", allow_frac=5e-3)
    assert_ulp(de, deo, dtype, ulps=U(dtype, 2), atol=at, what=f"{kind} bwd de", allow_frac=5e-3)


def test_glu_golden(golden):
    import unsloth_amd.kernels as K
    for dn, dt in (("f32", torch.float32), ("f16", torch.float16)):
        c = golden[f"glu_{dn}"]
        for kind, (f, b) in KINDS.items():
            h = getattr(K, f)(c["e"].to(DEV), c["g"].to(DEV))
            u = 8 if dt == torch.float32 else 1
            assert_ulp(h, c[kind + "_h"], dt, ulps=u, what=f"golden {kind} fwd {dn}", allow_frac=1e-2)
            out = getattr(K, b)(c["DW"].clone().to(DEV), c["e"].clone().view(10, 24).to(DEV),
                                c["g"].clone().view(10, 24).to(DEV))
            for a, w, nm in zip(out, c[kind + "_bwd"], ("h", "df", "de")):
                assert_ulp(a, w, dt, ulps=2 * u, what=f"golden {kind} bwd {nm} {dn}", allow_frac=1e-2)


# ---------------------------------------------------------------- cross entropy
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("V,softcap,scale", [(1000, 0, 0), (32000, 0, 0), (128256, 0, 0), (50257, 30.0, 0),
                                             (4099, 0, 0.125), (70001, 30.0, 0.5)])
def test_cross_entropy(dtype, V, softcap, scale):
    from unsloth_amd.kernels.cross_entropy_loss import Fast_CrossEntropyLoss
    rows = 12
    logits = (torch.randn(rows, V, generator=g(21)) * 4).to(dtype)
    labels = torch.randint(0, V, (rows,), generator=g(22))
    labels[3] = -100
    labels[7] = V - 1
    labels[8] = 0
    lo, lse = R.cross_entropy_forward(logits, labels, softcap, scale)
    lg = logits.to(DEV).requires_grad_(True)
    xin = lg * 1.0
    loss = Fast_CrossEntropyLoss.apply(xin, labels.to(DEV), softcap, scale)
    torch.testing.assert_close(loss.cpu(), lo, rtol=2e-5, atol=2e-5)
    assert loss[3].item() == 0.0
    dl = torch.rand(rows, generator=g(23))
    loss.backward(dl.to(DEV))
    want = R.cross_entropy_backward(logits, dl, lse, labels, softcap, scale)
    assert_ulp(lg.grad, want, dtype, ulps=U(dtype, 2), atol=1e-6 if dtype != torch.float32 else 1e-8, what="ce bwd",
               allow_frac=5e-3)
    assert torch.all(lg.grad[3] == 0), "ignored row must have an exactly zero gradient"
    # exactly one column per valid row carries the -1: gradient rows sum to ~0
    s = lg.grad.float().sum(dim=1).cpu()
    if not softcap and not scale and dtype == torch.float32:
        assert s.abs().max() < 1e-4


def test_cross_entropy_golden_and_mean(golden):
    from unsloth_amd.kernels.cross_entropy_loss import fast_cross_entropy_loss
    for dn, dt in (("f32", torch.float32), ("f16", torch.float16)):
        for tag in ("plain", "softcap", "scale"):
            c = golden[f"ce_{tag}_{dn}"]
            kw = {k: c[k] for k in ("logit_softcapping", "logit_scaling") if k in c}
            lg = c["logits"].to(DEV).requires_grad_(True)
            loss = fast_cross_entropy_loss(lg * 1.0, c["labels"].to(DEV), **kw)
            torch.testing.assert_close(loss.cpu().float(), c["loss"].float(), rtol=2e-5, atol=2e-6)
            loss.backward()
            assert_ulp(lg.grad, c["dlogits"], dt, ulps=16 if dt == torch.float32 else 2, atol=1e-7,
                       what=f"golden ce {tag} {dn}", allow_frac=1e-2)
    c = golden["ce_chunked_f32"]
    lg = c["logits"].to(torch.float32).to(DEV).requires_grad_(True)
    loss = fast_cross_entropy_loss(lg * 1.0, c["labels"].to(DEV))
    torch.testing.assert_close(loss.cpu(), c["loss"], rtol=2e-5, atol=2e-6)
    loss.backward()
    torch.testing.assert_close(lg.grad[0, 0, -64:].cpu(), c["dlogits_row0_tail"], rtol=1e-4, atol=1e-7)
    assert torch.all(lg.grad[0, 1] == 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(64, 4096), (5, 2048), (33, 1024), (257, 64)])
def test_add_rms_layernorm_equals_add_then_norm(dtype, rows, cols):
    """residual add fused into the norm == torch add followed by the (oracle-checked) norm kernels, bit for bit,
    forward (h and y) and backward (dX with the residual-path gradient added inside the kernel)."""
    from unsloth_amd.kernels.rms_layernorm import Fast_Add_RMS_Layernorm, Fast_RMS_Layernorm, add_rms_supported
    if not add_rms_supported(torch.empty(1, cols, dtype=dtype, device=DEV), torch.empty(cols, dtype=dtype, device=DEV)):
        pytest.skip("row too long for the register-resident kernel: fast_add_rms_layernorm takes the two-op path")
    g = torch.Generator().manual_seed(cols + rows)
    x = torch.randn(rows, cols, generator=g).to(dtype).to(DEV)
    res = torch.randn(rows, cols, generator=g).to(dtype).to(DEV)
    W = torch.rand(cols, generator=g).to(dtype).to(DEV)
    dh = torch.randn(rows, cols, generator=g).to(dtype).to(DEV)
    dy = torch.randn(rows, cols, generator=g).to(dtype).to(DEV)
    # reference: separate ops
    x1, r1 = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    h1 = r1 + x1
    y1 = Fast_RMS_Layernorm.apply(h1, W, 1e-5, False)
    torch.autograd.backward([h1, y1], [dh.clone(), dy.clone()])
    # fused
    x2, r2 = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    h2, y2 = Fast_Add_RMS_Layernorm.apply(x2, r2, W, 1e-5)
    assert torch.equal(h2, h1) and torch.equal(y2, y1)
    torch.autograd.backward([h2, y2], [dh.clone(), dy.clone()])
    assert torch.equal(x2.grad, x1.grad) and torch.equal(r2.grad, r1.grad)
    # h unused downstream (last layer): dH is None
    x3, r3 = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    _, y3 = Fast_Add_RMS_Layernorm.apply(x3, r3, W, 1e-5)
    y3.backward(dy.clone())
    x4 = x.clone().requires_grad_(True)
    Fast_RMS_Layernorm.apply(res + x4, W, 1e-5, False).backward(dy.clone())
    assert torch.equal(x3.grad, x4.grad)
