"""Generates tests/golden/ref_triton.pt by running the REFERENCE's own Triton kernels and
manual-autograd Functions (imported read-only from /root/reference) on the CPU under
TRITON_INTERPRET=1, following the stub-harness recipe of SURVEY.md section 10.

Run in the build container only (`python oracle/make_golden_from_reference.py`); the GPU box has
no /root/reference and only ever reads the committed fixture. fp32 and fp16 only: the Triton
interpreter has no bf16 (numpy), see SURVEY 8(c).
"""
import importlib
import logging
import os
import re
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"
os.environ["UNSLOTH_ALLOW_CPU"] = "1"
import torch  # noqa: E402
import triton  # noqa: E402
import triton.language as tl  # noqa: E402
from packaging.version import Version as _V  # noqa: E402

REF = "/root/reference/unsloth"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_triton.pt")


def mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


class Version(_V):
    def __init__(self, v):
        super().__init__(re.match(r"[0-9.]+", str(v)).group(0).rstrip("."))


def load_reference():
    mod("unsloth_zoo").__path__ = []
    mod("unsloth_zoo.utils", Version=Version)
    mod("unsloth_zoo.log", logger=logging.getLogger("zoo"))
    mod("unsloth_zoo.temporary_patches").__path__ = []
    mod("unsloth_zoo.temporary_patches.common",
        torch_compile=lambda *a, **k: a[0] if (a and callable(a[0])) else (lambda f: f))
    mod("unsloth_zoo.loss_utils", patch_loss_functions=lambda *a, **k: None,
        post_patch_loss_function=lambda m: m)
    mod("unsloth_zoo.patching_utils", patch_layernorm=lambda *a, **k: None)
    mod("unsloth").__path__ = [REF]
    mod("unsloth.kernels").__path__ = [REF + "/kernels"]
    ku = importlib.import_module("unsloth.kernels.utils")
    ku.is_cdna = lambda: False
    rms = importlib.import_module("unsloth.kernels.rms_layernorm")
    rope = importlib.import_module("unsloth.kernels.rope_embedding")
    ce = importlib.import_module("unsloth.kernels.cross_entropy_loss")
    ce.is_cdna = lambda: False

    @triton.jit
    def _tcast(x, dtype):
        return tl.cast(x, dtype)

    @triton.jit
    def _ttanh(x):
        return 2.0 * tl.sigmoid(2.0 * x) - 1.0

    ce.triton_cast = _tcast
    sw = importlib.import_module("unsloth.kernels.swiglu")
    ge = importlib.import_module("unsloth.kernels.geglu")
    ge.triton_tanh = _ttanh
    ce.triton_tanh = _ttanh
    fl = importlib.import_module("unsloth.kernels.fast_lora")
    return rms, rope, ce, sw, ge, fl


def main():
    rms, rope, ce, sw, ge, fl = load_reference()
    G = {}
    gen = torch.Generator().manual_seed(3407)
    rnd = lambda *s, dtype=torch.float32, scale=1.0: (torch.randn(*s, generator=gen) * scale).to(dtype)

    for dt_name, dt in (("f32", torch.float32), ("f16", torch.float16)):
        # ---- RMSNorm (rows 5, cols 96: not a power of two on purpose)
        for gemma in (False, True):
            X = rnd(5, 96, dtype=dt).requires_grad_(True)
            W = torch.rand(96, generator=gen).to(dt)
            Y = rms.Fast_RMS_Layernorm.apply(X, W, 1e-5, gemma)
            dY = rnd(5, 96, dtype=dt)
            dY_in = dY.clone()
            Y.backward(dY)
            G[f"rms_{dt_name}_gemma{int(gemma)}"] = dict(X=X.detach().clone(), W=W, eps=1e-5, Y=Y.detach().clone(),
                                                        dY=dY_in, dX=X.grad.clone())
        # ---- RoPE dense + QK with restarting indices
        B, H, Hk, T, D = 2, 4, 2, 6, 16
        pos = torch.arange(32, dtype=torch.float32)
        inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
        fr = torch.outer(pos, inv)
        emb = torch.cat((fr, fr), dim=-1)
        cos, sin = emb.cos().to(dt), emb.sin().to(dt)
        Q = rnd(B, H, T, D, dtype=dt)
        K = rnd(B, Hk, T, D, dtype=dt)
        idx = torch.tensor([0, 1, 2, 0, 1, 2, 0, 1, 0, 1, 2, 3], dtype=torch.int32)   # packed docs
        Qo, Ko = rope.fast_rope_embedding(Q.clone(), K.clone(), cos, sin, idx)
        Qd, Kd = rope.fast_rope_embedding(Q.clone(), K.clone(), cos, sin, None)
        dQ = rnd(B, H, T, D, dtype=dt)
        dK = rnd(B, Hk, T, D, dtype=dt)
        Qg, Kg = Q.clone().requires_grad_(True), K.clone().requires_grad_(True)
        # route through a multiply so the in-place kernel does not write into a leaf
        qo, ko = rope.Fast_RoPE_Embedding_QK.apply(Qg * 1.0, Kg * 1.0, cos, sin, idx)
        torch.autograd.backward([qo, ko], [dQ.clone(), dK.clone()])
        G[f"rope_{dt_name}"] = dict(Q=Q, K=K, cos=cos, sin=sin, idx=idx, Q_idx=Qo, K_idx=Ko, Q_dense=Qd,
                                    K_dense=Kd, dQ=dQ, dK=dK, dQ_in=Qg.grad.clone(), dK_in=Kg.grad.clone())
        # ---- GLU family
        e = rnd(2, 5, 24, dtype=dt)
        g = rnd(2, 5, 24, dtype=dt)
        DW = rnd(10, 24, dtype=dt)
        ent = dict(e=e, g=g, DW=DW)
        for name, f, b in (("swiglu", sw.swiglu_fg_kernel, sw.swiglu_DWf_DW_dfg_kernel),
                           ("geglu_exact", ge.geglu_exact_forward_kernel, ge.geglu_exact_backward_kernel),
                           ("geglu_approx", ge.geglu_approx_forward_kernel, ge.geglu_approx_backward_kernel)):
            ent[name + "_h"] = f(e.clone(), g.clone())
            h2, df, de = b(DW.clone(), e.clone().view(10, 24), g.clone().view(10, 24))
            ent[name + "_bwd"] = (h2.clone(), df.clone(), de.clone())
        G[f"glu_{dt_name}"] = ent
        # ---- cross entropy: single block, softcap, scaling
        for tag, V, kw in (("plain", 1000, {}), ("softcap", 500, dict(logit_softcapping=30.0)),
                           ("scale", 500, dict(logit_scaling=0.125))):
            logits = rnd(2, 4, V, dtype=dt, scale=4.0)
            labels = torch.randint(0, V, (2, 4), generator=gen)
            labels[0, 1] = -100
            lg = logits.clone().requires_grad_(True)
            loss = ce.fast_cross_entropy_loss(lg * 1.0, labels, **kw)
            loss.backward()
            G[f"ce_{tag}_{dt_name}"] = dict(logits=logits, labels=labels, loss=loss.detach().clone(),
                                            dlogits=lg.grad.clone(), **kw)
    # ---- chunked CE (V > 65536), fp32 only to bound the fixture size
    V = 70000
    logits = rnd(1, 2, V, scale=4.0)
    labels = torch.tensor([[69999, -100]])
    lg = logits.clone().requires_grad_(True)
    loss = ce.fast_cross_entropy_loss(lg * 1.0, labels)
    loss.backward()
    G["ce_chunked_f32"] = dict(logits=logits.to(torch.float16), labels=labels, loss=loss.detach().clone(),
                               dlogits_row0_tail=lg.grad[0, 0, -64:].clone(), note="logits stored as fp16 "
                               "(values exactly representable: generated then rounded)")
    # recompute with the rounded logits so the stored fixture is self-consistent
    logits = G["ce_chunked_f32"]["logits"].to(torch.float32)
    lg = logits.clone().requires_grad_(True)
    loss = ce.fast_cross_entropy_loss(lg * 1.0, labels)
    loss.backward()
    G["ce_chunked_f32"].update(loss=loss.detach().clone(), dlogits_row0_tail=lg.grad[0, 0, -64:].clone(),
                               dlogits_row1_absmax=lg.grad[0, 1].abs().max().clone())

    # ---- manual-autograd LoRA blocks on dense weights (W_quant=None), fp32
    dt = torch.float32
    Hd, I, r, Bz, T = 32, 48, 8, 2, 5
    X = rnd(Bz, T, Hd, scale=0.5)
    mk = lambda o, i: (rnd(o, i, scale=0.1), rnd(r, i, scale=0.1), rnd(o, r, scale=0.1), 2.0)
    gate, up, down = mk(I, Hd), mk(I, Hd), mk(Hd, I)
    leaves = []

    def leaf(t):
        t = t.clone().requires_grad_(True)
        leaves.append(t)
        return t

    Xg = leaf(X)
    gA, gB, uA, uB, dA, dB = (leaf(t) for t in (gate[1], gate[2], up[1], up[2], down[1], down[2]))
    out = fl.LoRA_MLP.apply(Xg * 1.0, gate[0], None, gA, gB, gate[3], up[0], None, uA, uB, up[3],
                            down[0], None, dA, dB, down[3], sw.swiglu_fg_kernel,
                            sw.swiglu_DWf_DW_dfg_kernel, False)
    dY = rnd(Bz, T, Hd)
    out.backward(dY)
    G["lora_mlp_f32"] = dict(X=X, gate=gate, up=up, down=down, dY=dY, out=out.detach().clone(),
                             grads=[t.grad.clone() for t in leaves])
    # QKV + O
    q, k, v = mk(Hd, Hd), mk(16, Hd), mk(16, Hd)
    leaves = []
    Xg = leaf(X)
    params = [leaf(t) for t in (q[1], q[2], k[1], k[2], v[1], v[2])]
    Qo, Ko, Vo = fl.LoRA_QKV.apply(Xg * 1.0, q[0], None, params[0], params[1], q[3], k[0], None, params[2],
                                   params[3], k[3], v[0], None, params[4], params[5], v[3], False)
    dQ, dK, dV = rnd(Bz, T, Hd), rnd(Bz, T, 16), rnd(Bz, T, 16)
    torch.autograd.backward([Qo, Ko, Vo], [dQ, dK, dV])
    G["lora_qkv_f32"] = dict(X=X, q=q, k=k, v=v, dQ=dQ, dK=dK, dV=dV, Q=Qo.detach().clone(),
                             K=Ko.detach().clone(), V=Vo.detach().clone(),
                             grads=[t.grad.clone() for t in leaves])
    o = mk(Hd, Hd)
    leaves = []
    Xg = leaf(X)
    oA, oB = leaf(o[1]), leaf(o[2])
    O = fl.LoRA_W.apply(Xg * 1.0, o[0], None, oA, oB, o[3])
    O.backward(dY)
    G["lora_w_f32"] = dict(X=X, o=o, dY=dY, out=O.detach().clone(), grads=[t.grad.clone() for t in leaves])

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    torch.save(G, OUT)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes;", len(G), "cases")


if __name__ == "__main__":
    main()
