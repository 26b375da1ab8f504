"""-m gpu: the HIP kernels against tests/golden/ref_triton_bf16.pt -- outputs of the REFERENCE's own Triton
kernels and manual-autograd Functions run natively in bf16 on an MI355X (oracle/make_golden_bf16_gpu.py over the
modules staged by oracle/stage_reference.py). This is the bf16 pin of rows a1-a10 / a12-a15: north_star asks for
"within 1e-3 bf16 of the reference's Triton path".

Tolerances, written out:
  * elementwise kernels (RMSNorm, RoPE, SwiGLU/GeGLU): every element within 1 bf16 ulp (2^-7 relative) of the
    reference, and at most `allow_frac` of the elements different at all by more than rounding noise -- both sides
    compute in fp32 and round once, so differences are single last-bit flips from exp/rsqrt implementations;
  * cross entropy: per-row loss (fp32) within 1e-3 relative / 1e-3 absolute of the reference (it reads bf16 logits),
    gradient within 2 bf16 ulp + 1e-6;
  * GEMM-carrying blocks (LoRA_MLP/QKV/W): the reference rounds the base product to bf16 BEFORE adding the LoRA
    term (utils.py:1158-1168) and runs hipBLASLt, ours adds in fp32 and rounds once, so the two bf16 pipelines
    differ by rounding noise (measured 5e-3 relative Frobenius on the MLP output). Criterion: against the fp32
    truth of the same bf16 inputs our error <= 1.25 x the reference's own error (floor 2e-3), and ours vs the
    reference <= 1.5e-2.
Every comparison also lands in gpurun_out/bf16_parity_report.json (max abs / max ulp / mismatch fraction).
"""
import json
import os

import pytest
import torch

from tests._util import assert_ulp, rel_fro

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "ref_triton_bf16.pt")
REPORT = {}


@pytest.fixture(scope="module")
def gold():
    if not os.path.isfile(FIXTURE):
        pytest.fail("tests/golden/ref_triton_bf16.pt is missing: run oracle/make_golden_bf16_gpu.py on the GPU box")
    G = torch.load(FIXTURE, weights_only=False)
    yield G
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bf16_parity_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def note(name, actual, expected):
    a, b = actual.detach().float().cpu(), expected.detach().float().cpu()
    err = (a - b).abs()
    ulp = err / (b.abs() * 2.0 ** -7 + 1e-30)
    REPORT[name] = dict(max_abs=float(err.max()), max_ulp=float(ulp[b.abs() > 1e-6].max()) if (b.abs() > 1e-6).any() else 0.0,
                        mismatch_frac=float((a != b).float().mean()), rel_fro=rel_fro(a, b), n=a.numel())


def test_fixture_is_native_bf16_and_complete(gold):
    meta = gold["_meta"]
    assert not meta["errors"], list(meta["errors"])
    assert meta["device_type"] == "hip" and "gfx950" in meta.get("arch", "gfx950"), meta      # native run on the MI355X
    fp16 = meta["native_vs_interpreter_fp16"]          # harness check: native fp16 == the CPU-interpreter fixture
    assert "error" not in fp16 and fp16["rms_f16_maxdiff"] == 0 and fp16["swiglu_f16_maxdiff"] == 0, fp16
    for k in ("rms_h4096_gemma0", "rope_llama3", "glu_i14336", "ce_v128256", "lora_mlp_mid", "lora_qkv_mid", "lora_w_mid"):
        assert k in gold, k


@pytest.mark.parametrize("tag", ["small", "h4096", "h2048"])
@pytest.mark.parametrize("gemma", [0, 1])
def test_rms_vs_reference_triton_bf16(gold, tag, gemma):
    from unsloth_amd.kernels.rms_layernorm import Fast_RMS_Layernorm
    c = gold[f"rms_{tag}_gemma{gemma}"]
    Xg = c["X"].to(DEV).requires_grad_(True)
    Y = Fast_RMS_Layernorm.apply(Xg, c["W"].to(DEV), c["eps"], bool(gemma))
    note(f"rms_{tag}_g{gemma}_Y", Y, c["Y"])
    assert_ulp(Y, c["Y"], BF, ulps=1, what="rms fwd vs ref-triton bf16", allow_frac=2e-3)
    Y.backward(c["dY"].to(DEV))
    note(f"rms_{tag}_g{gemma}_dX", Xg.grad, c["dX"])
    assert_ulp(Xg.grad, c["dX"], BF, ulps=2, what="rms bwd vs ref-triton bf16", allow_frac=5e-3)


@pytest.mark.parametrize("tag", ["small", "llama3"])
def test_rope_vs_reference_triton_bf16(gold, tag):
    from unsloth_amd.kernels.rope_embedding import Fast_RoPE_Embedding_QK, fast_rope_embedding
    c = gold[f"rope_{tag}"]
    cos, sin = c["cos"].to(DEV), c["sin"].to(DEV)
    for idx, qk, kk in ((c["idx"].to(DEV), "Q_idx", "K_idx"), (None, "Q_dense", "K_dense")):
        Qr, Kr = fast_rope_embedding(c["Q"].to(DEV).clone(), c["K"].to(DEV).clone(), cos, sin, idx)
        note(f"rope_{tag}_{qk}", Qr, c[qk])
        note(f"rope_{tag}_{kk}", Kr, c[kk])
        assert_ulp(Qr, c[qk], BF, ulps=1, atol=2.0 ** -9, what=f"rope {qk}", allow_frac=2e-3)
        assert_ulp(Kr, c[kk], BF, ulps=1, atol=2.0 ** -9, what=f"rope {kk}", allow_frac=2e-3)
    Qg, Kg = c["Q"].to(DEV).requires_grad_(True), c["K"].to(DEV).requires_grad_(True)
    qo, ko = Fast_RoPE_Embedding_QK.apply(Qg * 1.0, Kg * 1.0, cos, sin, c["idx"].to(DEV))
    torch.autograd.backward([qo, ko], [c["dQ"].to(DEV).clone(), c["dK"].to(DEV).clone()])
    note(f"rope_{tag}_dQ", Qg.grad, c["dQ_in"])
    assert_ulp(Qg.grad, c["dQ_in"], BF, ulps=1, atol=2.0 ** -9, what="rope dQ", allow_frac=2e-3)
    assert_ulp(Kg.grad, c["dK_in"], BF, ulps=1, atol=2.0 ** -9, what="rope dK", allow_frac=2e-3)


KINDS = {"swiglu": ("swiglu_fg_kernel", "swiglu_DWf_DW_dfg_kernel"),
         "geglu_exact": ("geglu_exact_forward_kernel", "geglu_exact_backward_kernel"),
         "geglu_approx": ("geglu_approx_forward_kernel", "geglu_approx_backward_kernel")}


@pytest.mark.parametrize("tag", ["small", "i14336"])
@pytest.mark.parametrize("kind", list(KINDS))
def test_glu_vs_reference_triton_bf16(gold, tag, kind):
    import unsloth_amd.kernels as K
    c = gold[f"glu_{tag}"]
    f, b = KINDS[kind]
    n = c["e"].shape[-1]
    h = getattr(K, f)(c["e"].to(DEV), c["g"].to(DEV))
    note(f"{kind}_{tag}_h", h, c[kind + "_h"])
    assert_ulp(h, c[kind + "_h"], BF, ulps=1, what=f"{kind} fwd", allow_frac=5e-3)
    out = getattr(K, b)(c["DW"].clone().to(DEV), c["e"].clone().view(-1, n).to(DEV), c["g"].clone().view(-1, n).to(DEV))
    for a, w, nm in zip(out, c[kind + "_bwd"], ("h", "df", "de")):
        note(f"{kind}_{tag}_bwd_{nm}", a, w)
        assert_ulp(a, w, BF, ulps=2, what=f"{kind} bwd {nm}", allow_frac=1e-2)


@pytest.mark.parametrize("tag", ["plain", "softcap", "scale", "v32000", "v128256", "v70000_softcap"])
def test_cross_entropy_vs_reference_triton_bf16(gold, tag):
    from unsloth_amd.kernels.cross_entropy_loss import fast_cross_entropy_loss
    c = gold[f"ce_{tag}"]
    kw = {k: c[k] for k in ("logit_softcapping", "logit_scaling") if k in c}
    lg = c["logits"].to(DEV).requires_grad_(True)
    loss = fast_cross_entropy_loss(lg * 1.0, c["labels"].to(DEV), **kw)
    note(f"ce_{tag}_loss", loss.reshape(1), c["loss"].reshape(1))
    torch.testing.assert_close(loss.cpu().float(), c["loss"].float(), rtol=1e-3, atol=1e-3)   # north_star bound
    torch.testing.assert_close(loss.cpu().float(), c["loss"].float(), rtol=2e-5, atol=2e-5)   # what we actually hold
    loss.backward()
    note(f"ce_{tag}_dlogits", lg.grad, c["dlogits"])
    assert_ulp(lg.grad, c["dlogits"], BF, ulps=2, atol=1e-6, what=f"ce {tag} dlogits", allow_frac=1e-2)
    ign = c["labels"] == -100
    assert torch.all(lg.grad.cpu()[ign] == 0)


def _dev(t):
    return t.to(DEV)


def _truth_linear(X, p):
    W, A, B, s = p
    return X @ W.float().t() + s * ((X @ A.float().t()) @ B.float().t())


def _closer_than_reference(name, ours, ref, truth, failures, slack=1.25, floor=2e-3, sanity=1.5e-2):
    """Two bf16 pipelines with different rounding points (the reference rounds X W^T to bf16 BEFORE adding the LoRA
    term and runs hipBLASLt; ours adds in fp32 and rounds once) cannot agree bit for bit. The criterion: measured
    against the fp32 truth computed from the same bf16 inputs, OUR error is no larger than the REFERENCE's own
    (x `slack`, with an absolute floor of 2e-3 = a quarter bf16 ulp), and the two agree within 1.5e-2."""
    e_ours, e_ref, e_pair = rel_fro(ours, truth), rel_fro(ref, truth), rel_fro(ours, ref)
    REPORT[name] = dict(err_ours_vs_fp32=e_ours, err_ref_vs_fp32=e_ref, ours_vs_ref=e_pair, n=truth.numel())
    if not (e_ours <= max(slack * e_ref, floor) and e_pair <= sanity):
        failures.append((name, e_ours, e_ref, e_pair))


@pytest.mark.parametrize("tag", ["small", "mid"])
def test_lora_blocks_vs_reference_bf16(gold, tag):
    """LoRA_MLP / LoRA_QKV / LoRA_W on dense bf16 weights: the reference composes hipBLASLt GEMMs + its Triton SwiGLU."""
    import unsloth_amd.kernels as K
    from unsloth_amd.kernels.fast_lora import LoRA_MLP, LoRA_QKV, LoRA_W
    failures = []
    # ---------------- MLP
    c = gold[f"lora_mlp_{tag}"]
    ins = (c["X"], c["gate"][1], c["gate"][2], c["up"][1], c["up"][2], c["down"][1], c["down"][2])
    leaves = [_dev(t).clone().requires_grad_(True) for t in ins]
    Xg, gA, gB, uA, uB, dA, dB = leaves
    out = LoRA_MLP.apply(Xg * 1.0, _dev(c["gate"][0]), None, gA, gB, c["gate"][3], _dev(c["up"][0]), None, uA, uB,
                         c["up"][3], _dev(c["down"][0]), None, dA, dB, c["down"][3], K.swiglu_fg_kernel,
                         K.swiglu_DWf_DW_dfg_kernel, False)
    out.backward(_dev(c["dY"]))
    t = [x.float().clone().requires_grad_(True) for x in ins]
    e = _truth_linear(t[0], (c["gate"][0], t[1], t[2], c["gate"][3]))
    g = _truth_linear(t[0], (c["up"][0], t[3], t[4], c["up"][3]))
    o32 = _truth_linear(torch.nn.functional.silu(e) * g, (c["down"][0], t[5], t[6], c["down"][3]))
    o32.backward(c["dY"].float())
    _closer_than_reference(f"lora_mlp_{tag}_out", out, c["out"], o32.detach(), failures)
    for i, (lf, w, tt) in enumerate(zip(leaves, c["grads"], t)):
        _closer_than_reference(f"lora_mlp_{tag}_grad{i}", lf.grad, w, tt.grad, failures)
    # ---------------- QKV
    c = gold[f"lora_qkv_{tag}"]
    q, k, v = c["q"], c["k"], c["v"]
    ins = (c["X"], q[1], q[2], k[1], k[2], v[1], v[2])
    leaves = [_dev(x).clone().requires_grad_(True) for x in ins]
    p = leaves[1:]
    Q, Kk, V = LoRA_QKV.apply(leaves[0] * 1.0, _dev(q[0]), None, p[0], p[1], q[3], _dev(k[0]), None, p[2], p[3], k[3],
                              _dev(v[0]), None, p[4], p[5], v[3], False)
    torch.autograd.backward([Q, Kk, V], [_dev(c["dQ"]), _dev(c["dK"]), _dev(c["dV"])])
    t = [x.float().clone().requires_grad_(True) for x in ins]
    Q32 = _truth_linear(t[0], (q[0], t[1], t[2], q[3]))
    K32 = _truth_linear(t[0], (k[0], t[3], t[4], k[3]))
    V32 = _truth_linear(t[0], (v[0], t[5], t[6], v[3]))
    torch.autograd.backward([Q32, K32, V32], [c["dQ"].float(), c["dK"].float(), c["dV"].float()])
    for nm, a, w, tr in (("Q", Q, c["Q"], Q32), ("K", Kk, c["K"], K32), ("V", V, c["V"], V32)):
        _closer_than_reference(f"lora_qkv_{tag}_{nm}", a, w, tr.detach(), failures)
    for i, (lf, w, tt) in enumerate(zip(leaves, c["grads"], t)):
        _closer_than_reference(f"lora_qkv_{tag}_grad{i}", lf.grad, w, tt.grad, failures)
    # ---------------- O
    c = gold[f"lora_w_{tag}"]
    o = c["o"]
    ins = (c["X"], o[1], o[2])
    leaves = [_dev(x).clone().requires_grad_(True) for x in ins]
    O = LoRA_W.apply(leaves[0] * 1.0, _dev(o[0]), None, leaves[1], leaves[2], o[3])
    O.backward(_dev(c["dY"]))
    t = [x.float().clone().requires_grad_(True) for x in ins]
    O32 = _truth_linear(t[0], (o[0], t[1], t[2], o[3]))
    O32.backward(c["dY"].float())
    _closer_than_reference(f"lora_w_{tag}_out", O, c["out"], O32.detach(), failures)
    for i, (lf, w, tt) in enumerate(zip(leaves, c["grads"], t)):
        _closer_than_reference(f"lora_w_{tag}_grad{i}", lf.grad, w, tt.grad, failures)
    assert not failures, failures
