"""Pins the CPU oracle (oracle/ref_ops.py) against outputs of the REFERENCE's own Triton kernels
and manual-autograd Functions, recorded by oracle/make_golden_from_reference.py under
TRITON_INTERPRET=1 into tests/golden/ref_triton.pt. fp32: 1e-5 relative; fp16: at most one fp16
ulp (exp/sigmoid implementations differ in the last fp32 bit before the rounding point)."""
import pytest
import torch

from oracle import ref_ops as R

DT = {"f32": torch.float32, "f16": torch.float16}


def close(a, b, dt):
    a, b = a.to(torch.float32), b.to(torch.float32)
    if dt == torch.float32:
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    else:
        # <= 1 ulp of fp16 relative to magnitude, + tiny absolute floor
        tol = (b.abs() * 2 ** -10).clamp_min(2 ** -14)
        bad = (a - b).abs() > tol
        assert not bad.any(), f"{int(bad.sum())} / {bad.numel()} beyond 1 ulp; max diff {(a-b).abs().max()}"


@pytest.mark.parametrize("dn", ["f32", "f16"])
@pytest.mark.parametrize("gemma", [0, 1])
def test_rms(golden, dn, gemma):
    g = golden[f"rms_{dn}_gemma{gemma}"]
    Y, r = R.rms_layernorm_forward(g["X"], g["W"], g["eps"], bool(gemma))
    close(Y, g["Y"], DT[dn])
    dX = R.rms_layernorm_backward(g["dY"], g["X"], g["W"], r, bool(gemma))
    close(dX, g["dX"], DT[dn])


@pytest.mark.parametrize("dn", ["f32", "f16"])
def test_rope(golden, dn):
    g = golden[f"rope_{dn}"]
    Q, K = R.rope_embedding_qk(g["Q"], g["K"], g["cos"], g["sin"], g["idx"])
    close(Q, g["Q_idx"], DT[dn]); close(K, g["K_idx"], DT[dn])
    Q, K = R.rope_embedding_qk(g["Q"], g["K"], g["cos"], g["sin"], None)
    close(Q, g["Q_dense"], DT[dn]); close(K, g["K_dense"], DT[dn])
    # dense [B,T,H,D] formulation is the same numbers
    Qd = R.rope_embedding_dense(g["Q"].permute(0, 2, 1, 3).contiguous(), g["cos"], g["sin"])
    close(Qd.permute(0, 2, 1, 3), g["Q_dense"], DT[dn])
    dQ, dK = R.rope_embedding_qk(g["dQ"], g["dK"], g["cos"], g["sin"], g["idx"], backward=True)
    close(dQ, g["dQ_in"], DT[dn]); close(dK, g["dK_in"], DT[dn])


@pytest.mark.parametrize("dn", ["f32", "f16"])
@pytest.mark.parametrize("kind", ["swiglu", "geglu_exact", "geglu_approx"])
def test_glu(golden, dn, kind):
    g = golden[f"glu_{dn}"]
    close(R.glu_forward(g["e"], g["g"], kind), g[kind + "_h"], DT[dn])
    h, df, de = R.glu_backward(g["DW"], g["e"].view(10, 24), g["g"].view(10, 24), kind)
    for a, b in zip((h, df, de), g[kind + "_bwd"]):
        close(a, b, DT[dn])


@pytest.mark.parametrize("dn", ["f32", "f16"])
@pytest.mark.parametrize("tag", ["plain", "softcap", "scale"])
def test_ce(golden, dn, tag):
    g = golden[f"ce_{tag}_{dn}"]
    sc, ls = g.get("logit_softcapping", 0.0), g.get("logit_scaling", 0.0)
    logits, labels = g["logits"], g["labels"]
    loss = R.fast_cross_entropy_loss(logits, labels, sc, ls)
    torch.testing.assert_close(loss.float(), g["loss"].float(), rtol=1e-5, atol=1e-6)
    B, T, V = logits.shape
    rows, lse = R.cross_entropy_forward(logits.view(-1, V), labels.view(-1), sc, ls)
    n = torch.count_nonzero(labels != -100)
    dl = torch.full((B * T,), 1.0) / n
    d = R.cross_entropy_backward(logits.view(-1, V), dl, lse, labels.view(-1), sc, ls).view(B, T, V)
    close(d, g["dlogits"], DT[dn])
    assert torch.all(d[0, 1] == 0)          # label -100 row: exactly zero gradient


def test_ce_chunked_vocab(golden):
    g = golden["ce_chunked_f32"]
    logits = g["logits"].to(torch.float32)
    labels = g["labels"]
    V = logits.shape[-1]
    assert V > 65536
    loss = R.fast_cross_entropy_loss(logits, labels)
    torch.testing.assert_close(loss, g["loss"], rtol=1e-5, atol=1e-6)
    rows, lse = R.cross_entropy_forward(logits.view(-1, V), labels.view(-1))
    d = R.cross_entropy_backward(logits.view(-1, V), torch.ones(2), lse, labels.view(-1))
    torch.testing.assert_close(d[0, -64:], g["dlogits_row0_tail"], rtol=2e-5, atol=1e-7)
    assert float(g["dlogits_row1_absmax"]) == 0.0 and torch.all(d[1] == 0)


def test_lora_mlp(golden):
    g = golden["lora_mlp_f32"]
    out, e, gg, h = R.lora_mlp_forward(g["X"], g["gate"], g["up"], g["down"])
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    _, grads = R.lora_mlp_reference_grads(g["X"], g["gate"], g["up"], g["down"], g["dY"])
    for a, b in zip(grads, g["grads"]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


def test_lora_qkv_and_w(golden):
    g = golden["lora_qkv_f32"]
    X = g["X"]
    for name, dname, out in (("q", "dQ", "Q"), ("k", "dK", "K"), ("v", "dV", "V")):
        torch.testing.assert_close(R.matmul_lora(X, *g[name]), g[out], rtol=1e-4, atol=1e-5)
    dX = 0
    grads = []
    for name, dname in (("q", "dQ"), ("k", "dK"), ("v", "dV")):
        W, A, B, s = g[name]
        d, dA, dB = R.lora_linear_grads(X, g[dname], W, A, B, s)
        dX = dX + d
        grads += [dA, dB]
    for a, b in zip([dX] + grads, g["grads"]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    g = golden["lora_w_f32"]
    W, A, B, s = g["o"]
    torch.testing.assert_close(R.matmul_lora(g["X"], W, A, B, s), g["out"], rtol=1e-4, atol=1e-5)
    d, dA, dB = R.lora_linear_grads(g["X"], g["dY"], W, A, B, s)
    for a, b in zip((d, dA, dB), g["grads"]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
