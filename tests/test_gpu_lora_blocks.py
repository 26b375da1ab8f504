"""-m gpu: the manual-autograd LoRA blocks (LoRA_MLP / LoRA_QKV / LoRA_W) and the fused linear-CE,
through the HIP path, against the oracle (forward, with the reference's rounding points) and
against fp32 autograd (gradients)."""
import pytest
import torch

from oracle import ref_ops as R
from tests._util import rel_fro

pytestmark = pytest.mark.gpu
DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def _mk(out_f, in_f, r, seed, quant):
    from unsloth_amd.nf4 import quantize_nf4
    W = (torch.randn(out_f, in_f, generator=g(seed)) * 0.03).to(torch.bfloat16)
    A = (torch.randn(r, in_f, generator=g(seed + 1)) * 0.05)
    B = (torch.randn(out_f, r, generator=g(seed + 2)) * 0.05)
    s = 2.0
    if quant:
        packed, qs = quantize_nf4(W.to(DEV), compress_statistics=True)
        Wd = R.nf4_dequantize_state(packed, qs)            # what the GPU path multiplies by
        dev = (packed, qs)
    else:
        Wd, dev = W, (W.to(DEV), None)
    return dict(cpu=(Wd, A, B, s), dev=dev, A=A.to(DEV).requires_grad_(True), B=B.to(DEV).requires_grad_(True), s=s)


@pytest.mark.parametrize("quant", [True, False])
@pytest.mark.parametrize("kind", ["swiglu", "geglu_exact", "geglu_approx"])
def test_lora_mlp(quant, kind):
    import unsloth_amd.kernels as K
    from unsloth_amd.kernels.fast_lora import LoRA_MLP
    H, I, r, Bz, T = 256, 704, 16, 2, 75
    gate, up, down = _mk(I, H, r, 1, quant), _mk(I, H, r, 11, quant), _mk(H, I, r, 21, quant)
    X = (torch.randn(Bz, T, H, generator=g(5)) * 0.5).to(torch.bfloat16)
    dY = torch.randn(Bz, T, H, generator=g(6)).to(torch.bfloat16)
    fwd = {"swiglu": K.swiglu_fg_kernel, "geglu_exact": K.geglu_exact_forward_kernel,
           "geglu_approx": K.geglu_approx_forward_kernel}[kind]
    bwd = {"swiglu": K.swiglu_DWf_DW_dfg_kernel, "geglu_exact": K.geglu_exact_backward_kernel,
           "geglu_approx": K.geglu_approx_backward_kernel}[kind]
    Xg = X.to(DEV).requires_grad_(True)
    out = LoRA_MLP.apply(Xg * 1.0, gate["dev"][0], gate["dev"][1], gate["A"], gate["B"], gate["s"],
                         up["dev"][0], up["dev"][1], up["A"], up["B"], up["s"],
                         down["dev"][0], down["dev"][1], down["A"], down["B"], down["s"], fwd, bwd, True)
    want, _, _, _ = R.lora_mlp_forward(X, gate["cpu"], up["cpu"], down["cpu"], kind)
    assert rel_fro(out, want) < 6e-3, rel_fro(out, want)
    out.backward(dY.to(DEV))
    _, grads = R.lora_mlp_reference_grads(X, gate["cpu"], up["cpu"], down["cpu"], dY, kind)
    got = [Xg.grad, gate["A"].grad, gate["B"].grad, up["A"].grad, up["B"].grad, down["A"].grad, down["B"].grad]
    names = ["dX", "d_gateA", "d_gateB", "d_upA", "d_upB", "d_downA", "d_downB"]
    for nm, a, b in zip(names, got, grads):
        assert a is not None, nm
        assert a.dtype == (torch.bfloat16 if nm == "dX" else torch.float32)
        assert rel_fro(a, b) < 2e-2, (nm, rel_fro(a, b))


@pytest.mark.parametrize("quant", [True, False])
def test_lora_qkv_and_o(quant):
    from unsloth_amd.kernels.fast_lora import LoRA_QKV, LoRA_W
    H, Hkv, r, Bz, T = 512, 128, 16, 1, 130
    q, k, v, o = _mk(H, H, r, 31, quant), _mk(Hkv, H, r, 41, quant), _mk(Hkv, H, r, 51, quant), _mk(H, H, r, 61, quant)
    X = (torch.randn(Bz, T, H, generator=g(7)) * 0.5).to(torch.bfloat16)
    Xg = X.to(DEV).requires_grad_(True)
    Q, Kk, V = LoRA_QKV.apply(Xg * 1.0, q["dev"][0], q["dev"][1], q["A"], q["B"], q["s"],
                              k["dev"][0], k["dev"][1], k["A"], k["B"], k["s"],
                              v["dev"][0], v["dev"][1], v["A"], v["B"], v["s"], True)
    for got, p in ((Q, q), (Kk, k), (V, v)):
        assert rel_fro(got, R.matmul_lora(X, *p["cpu"])) < 5e-3
    dQ = torch.randn(Bz, T, H, generator=g(8)).to(torch.bfloat16)
    dK = torch.randn(Bz, T, Hkv, generator=g(9)).to(torch.bfloat16)
    dV = torch.randn(Bz, T, Hkv, generator=g(10)).to(torch.bfloat16)
    torch.autograd.backward([Q, Kk, V], [dQ.to(DEV), dK.to(DEV), dV.to(DEV)])
    dX = 0
    for p, d in ((q, dQ), (k, dK), (v, dV)):
        W, A, B, s = p["cpu"]
        dx, dA, dB = R.lora_linear_grads(X, d, W, A, B, s)
        dX = dX + dx
        assert rel_fro(p["A"].grad, dA) < 2e-2 and rel_fro(p["B"].grad, dB) < 2e-2
    assert rel_fro(Xg.grad, dX) < 1e-2, rel_fro(Xg.grad, dX)
    # o_proj
    Xg2 = X.to(DEV).requires_grad_(True)
    O = LoRA_W.apply(Xg2 * 1.0, o["dev"][0], o["dev"][1], o["A"], o["B"], o["s"])
    assert rel_fro(O, R.matmul_lora(X, *o["cpu"])) < 5e-3
    O.backward(dQ.to(DEV))
    dx, dA, dB = R.lora_linear_grads(X, dQ, *o["cpu"])
    assert rel_fro(Xg2.grad, dx) < 1e-2 and rel_fro(o["A"].grad, dA) < 2e-2 and rel_fro(o["B"].grad, dB) < 2e-2


def test_lora_qkv_backward_merged_gemm(monkeypatch):
    """dQ | dK | dV arriving as column blocks of ONE buffer (what the flash-attention backward produces): dX runs as a
    single K-concatenated GEMM with the three rank-r terms side by side. Same oracle, same tolerances; the merged
    path must actually be taken (one GEMM launch for dX instead of three)."""
    from unsloth_amd.kernels import utils as U
    from unsloth_amd.kernels.fast_lora import LoRA_QKV
    H, Hkv, r, Bz, T = 512, 128, 16, 2, 96
    q, k, v = _mk(H, H, r, 32, True), _mk(Hkv, H, r, 42, True), _mk(Hkv, H, r, 52, True)
    X = (torch.randn(Bz, T, H, generator=g(17)) * 0.5).to(torch.bfloat16)
    Xg = X.to(DEV).requires_grad_(True)
    Q, Kk, V = LoRA_QKV.apply(Xg * 1.0, q["dev"][0], q["dev"][1], q["A"], q["B"], q["s"],
                              k["dev"][0], k["dev"][1], k["A"], k["B"], k["s"],
                              v["dev"][0], v["dev"][1], v["A"], v["B"], v["s"], True)
    d = torch.randn(Bz, T, H + 2 * Hkv, generator=g(18)).to(torch.bfloat16)
    dd = d.to(DEV)
    launches = []
    orig = U._launch_gemm
    monkeypatch.setattr(U, "_launch_gemm", lambda X2d, groups, nf4, accumulate=False, nn=False: (
        launches.append((tuple(X2d.shape), accumulate)), orig(X2d, groups, nf4, accumulate, nn))[1])
    torch.autograd.backward([Q, Kk, V], [dd[..., :H], dd[..., H:H + Hkv], dd[..., H + Hkv:]])
    assert launches == [((Bz * T, H + 2 * Hkv), False)], launches
    dX = 0
    for p, dy in ((q, d[..., :H]), (k, d[..., H:H + Hkv]), (v, d[..., H + Hkv:])):
        W, A, B, s = p["cpu"]
        dx, dA, dB = R.lora_linear_grads(X, dy, W, A, B, s)
        dX = dX + dx
        assert rel_fro(p["A"].grad, dA) < 2e-2 and rel_fro(p["B"].grad, dB) < 2e-2
    assert rel_fro(Xg.grad, dX) < 1e-2, rel_fro(Xg.grad, dX)


def test_matmul_lora_reference_signature():
    """matmul_lora(X, W, W_quant, A, B, s) incl. the transposed-weight call of the backward
    (fast_lora.py:156: matmul_lora(dY, W.t(), q, B.t(), A.t(), s) == dY @ W + s (dY B) A)."""
    from unsloth_amd.kernels import matmul_lora
    p = _mk(384, 256, 16, 71, True)
    W, A, B, s = p["cpu"]
    X = (torch.randn(3, 40, 256, generator=g(11)) * 0.5).to(torch.bfloat16)
    packed, qs = p["dev"]
    y = matmul_lora(X.to(DEV), packed, qs, p["A"].detach(), p["B"].detach(), s)
    assert y.shape == (3, 40, 384) and rel_fro(y, R.matmul_lora(X, W, A, B, s)) < 5e-3
    dY = torch.randn(120, 384, generator=g(12)).to(torch.bfloat16)
    At, Bt = p["A"].detach().to(torch.bfloat16).t(), p["B"].detach().to(torch.bfloat16).t()
    dx = matmul_lora(dY.to(DEV), packed.t(), qs, Bt, At, s)
    want, _, _ = R.lora_linear_grads(torch.zeros(120, 256), dY, W, A, B, s)
    assert dx.shape == (120, 256) and rel_fro(dx, want) < 1e-2


@pytest.mark.parametrize("V,softcap", [(1000, 0.0), (32000, 0.0), (128256, 0.0), (5003 * 8, 30.0),
                                       (32001, 0.0), (1003, 30.0)])       # vocab % 8 != 0: one added pad token
def test_fused_linear_ce(V, softcap):
    from unsloth_amd.kernels import unsloth_fused_ce_loss
    B, T, H = 2, 96, 256
    hidden = (torch.randn(B, T, H, generator=g(13)) * 0.7).to(torch.bfloat16)
    Wt = (torch.randn(V, H, generator=g(14)) * 0.05).to(torch.bfloat16)
    labels = torch.randint(0, V, (B, T), generator=g(15))
    labels[0, 5] = -100
    labels[1, :10] = -100
    want, dh_want = R.fused_linear_ce(hidden, Wt, labels, None, softcap)
    hg = hidden.to(DEV).requires_grad_(True)
    loss = unsloth_fused_ce_loss(None, hg * 1.0, Wt.to(DEV), None, labels.to(DEV), logit_softcapping=softcap,
                                 chunk_rows=64)
    torch.testing.assert_close(loss.cpu().float(), want.float(), rtol=1e-3, atol=1e-3)   # north star: 1e-3 bf16
    (loss / 3.0).backward()      # an upstream scale that is NOT a bf16 number: applied in fp32, rounded once
    assert rel_fro(hg.grad, dh_want.float() / 3.0) < 2e-2, rel_fro(hg.grad, dh_want.float() / 3.0)
    # n_items given (global token count under DP) only rescales
    loss2 = unsloth_fused_ce_loss(None, hidden.to(DEV), Wt.to(DEV), None, labels.to(DEV), n_items=1000,
                                  logit_softcapping=softcap)
    n = torch.count_nonzero(R.shift_labels(labels) != -100)
    torch.testing.assert_close(loss2.cpu().float() * 1000 / n, want.float(), rtol=1e-3, atol=1e-3)


def test_fused_ce_chunk_rows_policy():
    """chunk sizing (f3): `target_gb` bounds the transient [rows, V] logits chunk, multiples of 256 rows, 4096 max."""
    from unsloth_amd.kernels.cross_entropy_loss import fused_ce_chunk_rows
    dev = torch.device(DEV)
    assert fused_ce_chunk_rows(8192, 128256, 2, dev, target_gb=1.0) == 4096          # 1.05 GB chunk > 1 GiB? 4096*128256*2 = 0.98 GiB
    assert fused_ce_chunk_rows(8192, 128256, 2, dev, target_gb=0.25) == 1024
    assert fused_ce_chunk_rows(8192, 128256, 2, dev, target_gb=0.01) == 256
    assert fused_ce_chunk_rows(300, 32000, 2, dev, target_gb=4.0) == 512             # never more rows than the batch has
    assert fused_ce_chunk_rows(8192, 128256, 2, dev) in range(256, 4097, 256)
