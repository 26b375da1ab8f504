"""TEST INFRASTRUCTURE (oracle/): bf16 golden vectors from the REFERENCE's own Triton kernels, run natively.

Runs ON THE GPU BOX (triton-rocm on the MI355X): imports the reference modules staged by
`oracle/stage_reference.py` under oracle/_ref/unsloth (or /root/reference/unsloth when that exists) through the
stub harness of SURVEY.md section 10 -- WITHOUT the interpreter -- and executes them in bf16, the benchmark
dtype that `oracle/make_golden_from_reference.py` (CPU interpreter, fp32/fp16 only) cannot cover.

    gpurun -- 'python oracle/make_golden_bf16_gpu.py --out gpurun_out/ref_triton_bf16.pt'
    cp gpurun_out/ref_triton_bf16.pt tests/golden/          # committed fixture

Cases: RMSNorm (+gemma) fwd/bwd, RoPE Q+K indexed/dense fwd/bwd, SwiGLU / GeGLU exact+approx fwd/bwd,
cross entropy plain / softcap / scaled / V=128256 (chunked path) fwd/bwd, and the manual-autograd
LoRA_MLP / LoRA_QKV / LoRA_W Functions on dense bf16 weights -- each at a small shape and at Llama-3-8B widths
(hidden 4096, intermediate 14336, head_dim 128, vocab 128256). Every case stores its inputs, so the parity tests
(tests/test_gpu_ref_bf16_golden.py) feed the HIP kernels the very same bits.
It also re-runs the fp16 cases of the interpreter fixture natively and records the differences (harness check).

`--only linear_ce --out gpurun_out/ref_triton_bf16_linear_ce.pt` writes the SEPARATE fixture that pins row a7 (fused
linear cross entropy): the reference's own materialised-logits branch, unsloth/models/llama.py:1525-1562 --
`logits = lm_head(hidden)` in bf16, the label shift of :1545-1551, `fast_cross_entropy_loss` (the native Triton kernels of
unsloth/kernels/cross_entropy_loss.py:421-449) and autograd back to `d hidden` -- which SURVEY 8(c) and the reference's own
comment at llama.py:1490-1496 declare equivalent to the third-party `unsloth_fused_ce_loss` our kernel replaces. The lm_head
weight is stored as (seed, scale, shape, crc32) and rebuilt by `linear_ce_weight` (torch's CPU generator is deterministic),
so the fixture stays small at V = 128256.
"""
import argparse
import importlib
import logging
import os
import re
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _find_reference():
    for root in (os.path.join(HERE, "_ref", "unsloth"), "/root/reference/unsloth"):
        if os.path.isfile(os.path.join(root, "kernels", "rms_layernorm.py")):
            return root
    raise SystemExit("no staged reference: run `python oracle/stage_reference.py` in the build container first")


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def load_reference(ref):
    from packaging.version import Version as _V

    class Version(_V):
        def __init__(self, v):
            super().__init__(re.match(r"[0-9.]+", str(v)).group(0).rstrip("."))

    _mod("unsloth_zoo").__path__ = []
    _mod("unsloth_zoo.utils", Version=Version)
    _mod("unsloth_zoo.log", logger=logging.getLogger("zoo"))
    _mod("unsloth_zoo.temporary_patches").__path__ = []
    _mod("unsloth_zoo.temporary_patches.common",
         torch_compile=lambda *a, **k: a[0] if (a and callable(a[0])) else (lambda f: f))
    _mod("unsloth_zoo.loss_utils", patch_loss_functions=lambda *a, **k: None, post_patch_loss_function=lambda m: m)
    _mod("unsloth_zoo.patching_utils", patch_layernorm=lambda *a, **k: None)
    _mod("unsloth").__path__ = [ref]
    _mod("unsloth.kernels").__path__ = [os.path.join(ref, "kernels")]
    names = ("utils", "rms_layernorm", "rope_embedding", "cross_entropy_loss", "swiglu", "geglu", "fast_lora")
    return {n: importlib.import_module("unsloth.kernels." + n) for n in names}


def linear_ce_weight(seed, V, H, scale):
    """The lm_head weight of a linear_ce case, rebuilt from its seed (bf16 [V, H]) + the crc32 of its bits."""
    import zlib
    W = (torch.randn(V, H, generator=torch.Generator().manual_seed(seed)) * scale).to(torch.bfloat16)
    return W, zlib.crc32(W.view(torch.int16).numpy().tobytes())


LINEAR_CE_CASES = (
    # tag, V, H, B, T, kwargs of fast_cross_entropy_loss, n_items, fraction of ignored labels
    ("v32000", 32000, 256, 2, 48, {}, None, 0.1),
    ("v128256", 128256, 128, 1, 64, {}, None, 0.1),
    ("v128256_nitems", 128256, 128, 2, 40, {}, 1000, 0.5),
    ("v32000_softcap", 32000, 256, 2, 48, dict(logit_softcapping=30.0), None, 0.1),
    ("v70000_scale", 70000, 128, 1, 56, dict(logit_scaling=0.125), 333, 0.2),
    ("v32001_odd", 32001, 256, 1, 72, {}, None, 0.0),
)


def linear_ce_cases(ce, dev):
    """llama.py:1525-1562 run as the reference runs it (see the module docstring). Returns {name: case}."""
    G = {}
    for i, (tag, V, H, B, T, kw, n_items, ignore) in enumerate(LINEAR_CE_CASES):
        gen = torch.Generator().manual_seed(9000 + i)
        hidden = (torch.randn(B, T, H, generator=gen) * 0.7).to(torch.bfloat16)
        labels = torch.randint(0, V, (B, T), generator=gen)
        labels[torch.rand(B, T, generator=gen) < ignore] = -100
        labels[0, 1] = V - 1
        W, crc = linear_ce_weight(7000 + i, V, H, 0.05)
        hg = hidden.to(dev).requires_grad_(True)
        Wd = W.to(dev)
        logits = torch.nn.functional.linear(hg, Wd)                     # self.lm_head(hidden_states.to(dtype))   :1525
        lab = labels.to(dev)
        shift = torch.empty_like(lab)                                   # :1545-1551
        shift[..., :-1] = lab[..., 1:]
        shift[..., -1] = -100
        loss = ce.fast_cross_entropy_loss(logits=logits, labels=shift, n_items=n_items, **kw)      # :1556-1562
        loss.backward()
        G["linear_ce_" + tag] = dict(hidden=hidden, labels=labels, W_seed=7000 + i, W_scale=0.05, V=V, H=H, W_crc32=crc,
                                     n_items=n_items, kw=dict(kw), loss=loss.detach().float().cpu().clone(),
                                     dhidden=hg.grad.detach().cpu().clone())
        print("ok   linear_ce_" + tag, float(loss.detach()), flush=True)
    return G


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "gpurun_out", "ref_triton_bf16.pt"))
    ap.add_argument("--only", default="", help="'linear_ce': write only the a7 fixture (see the module docstring)")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs the MI355X (native Triton run)")
    import triton
    ref = _find_reference()
    R = load_reference(ref)
    rms, rope, ce, sw, ge, fl, ku = (R["rms_layernorm"], R["rope_embedding"], R["cross_entropy_loss"], R["swiglu"],
                                     R["geglu"], R["fast_lora"], R["utils"])
    dev = "cuda"
    if a.only == "linear_ce":
        G = linear_ce_cases(ce, dev)
        G["_meta"] = dict(torch=torch.__version__, triton=triton.__version__, device=torch.cuda.get_device_name(0),
                          arch=getattr(torch.cuda.get_device_properties(0), "gcnArchName", "?"),
                          device_type=getattr(ku, "DEVICE_TYPE", None), reference_root=ref,
                          note="llama.py:1525-1562: lm_head -> label shift -> fast_cross_entropy_loss (native Triton) -> d hidden")
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        torch.save(G, a.out)
        print("wrote", os.path.abspath(a.out), os.path.getsize(a.out), "bytes;", len(G) - 1, "cases")
        return
    G, errors = {}, {}
    gen = torch.Generator().manual_seed(3407)

    def rnd(*s, dtype=torch.bfloat16, scale=1.0):
        return (torch.randn(*s, generator=gen) * scale).to(dtype)

    def cpu(t):
        return t.detach().to("cpu").clone()

    def case(name):
        def deco(f):
            try:
                G[name] = f()
                print("ok  ", name, flush=True)
            except Exception as e:      # keep going: one broken case must not cost the GPU call
                import traceback
                errors[name] = traceback.format_exc()
                print("FAIL", name, repr(e), flush=True)
            return f
        return deco

    bf = torch.bfloat16
    # ------------------------------------------------------------------ RMSNorm
    for tag, rows, cols in (("small", 5, 96), ("h4096", 8, 4096), ("h2048", 3, 2048)):
        for gemma in (False, True):
            @case(f"rms_{tag}_gemma{int(gemma)}")
            def _():
                X = rnd(rows, cols)
                W = torch.rand(cols, generator=gen).to(bf)
                dY = rnd(rows, cols)
                Xg = X.to(dev).requires_grad_(True)
                Y = rms.Fast_RMS_Layernorm.apply(Xg, W.to(dev), 1e-5, gemma)
                Y.backward(dY.to(dev).clone())
                return dict(X=X, W=W, eps=1e-5, gemma=gemma, Y=cpu(Y), dY=dY, dX=cpu(Xg.grad))

    # ------------------------------------------------------------------ RoPE
    def tables(T, D, theta):
        inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
        fr = torch.outer(torch.arange(T, dtype=torch.int64).float(), inv)
        emb = torch.cat((fr, fr), dim=-1)
        return emb.cos().to(bf), emb.sin().to(bf)

    for tag, B, H, Hk, T, D, theta in (("small", 2, 4, 2, 6, 16, 1e4), ("llama3", 1, 32, 8, 12, 128, 5e5)):
        @case(f"rope_{tag}")
        def _():
            cos, sin = tables(64, D, theta)
            Q, K = rnd(B, H, T, D), rnd(B, Hk, T, D)
            if tag == "small":
                idx = torch.tensor([0, 1, 2, 0, 1, 2, 0, 1, 0, 1, 2, 3], dtype=torch.int32)
            else:
                idx = torch.cat([torch.arange(5), torch.arange(4), torch.arange(40, 43)]).to(torch.int32)
            cd, sd = cos.to(dev), sin.to(dev)
            Qo, Ko = rope.fast_rope_embedding(Q.to(dev).clone(), K.to(dev).clone(), cd, sd, idx.to(dev))
            Qd, Kd = rope.fast_rope_embedding(Q.to(dev).clone(), K.to(dev).clone(), cd, sd, None)
            dQ, dK = rnd(B, H, T, D), rnd(B, Hk, T, D)
            Qg, Kg = Q.to(dev).requires_grad_(True), K.to(dev).requires_grad_(True)
            qo, ko = rope.Fast_RoPE_Embedding_QK.apply(Qg * 1.0, Kg * 1.0, cd, sd, idx.to(dev))
            torch.autograd.backward([qo, ko], [dQ.to(dev).clone(), dK.to(dev).clone()])
            return dict(Q=Q, K=K, cos=cos, sin=sin, idx=idx, Q_idx=cpu(Qo), K_idx=cpu(Ko), Q_dense=cpu(Qd),
                        K_dense=cpu(Kd), dQ=dQ, dK=dK, dQ_in=cpu(Qg.grad), dK_in=cpu(Kg.grad))

    # ------------------------------------------------------------------ GLU family
    for tag, b, t, n in (("small", 2, 5, 24), ("i14336", 1, 2, 14336)):
        @case(f"glu_{tag}")
        def _():
            e, g = rnd(b, t, n), rnd(b, t, n)
            DW = rnd(b * t, n)
            ent = dict(e=e, g=g, DW=DW)
            for name, f, bw in (("swiglu", sw.swiglu_fg_kernel, sw.swiglu_DWf_DW_dfg_kernel),
                                ("geglu_exact", ge.geglu_exact_forward_kernel, ge.geglu_exact_backward_kernel),
                                ("geglu_approx", ge.geglu_approx_forward_kernel, ge.geglu_approx_backward_kernel)):
                ent[name + "_h"] = cpu(f(e.to(dev).clone(), g.to(dev).clone()))
                h2, df, de = bw(DW.to(dev).clone(), e.to(dev).clone().view(b * t, n), g.to(dev).clone().view(b * t, n))
                ent[name + "_bwd"] = (cpu(h2), cpu(df), cpu(de))
            return ent

    # ------------------------------------------------------------------ cross entropy
    for tag, V, kw in (("plain", 1000, {}), ("softcap", 500, dict(logit_softcapping=30.0)),
                       ("scale", 500, dict(logit_scaling=0.125)), ("v32000", 32000, {}), ("v128256", 128256, {}),
                       ("v70000_softcap", 70000, dict(logit_softcapping=30.0))):
        @case(f"ce_{tag}")
        def _():
            Bc, Tc = (2, 4) if V <= 1000 else ((1, 4) if V <= 32000 else (1, 3 if V < 100000 else 2))
            logits = rnd(Bc, Tc, V, scale=4.0)
            labels = torch.randint(0, V, (Bc, Tc), generator=gen)
            labels[0, 0 if Tc == 2 else 1] = -100
            if V > 1000:
                labels[0, Tc - 1] = V - 1
            lg = logits.to(dev).requires_grad_(True)
            loss = ce.fast_cross_entropy_loss(lg * 1.0, labels.to(dev), **kw)
            loss.backward()
            return dict(logits=logits, labels=labels, loss=cpu(loss), dlogits=cpu(lg.grad), **kw)

    # ------------------------------------------------------------------ manual-autograd LoRA blocks (dense bf16 W)
    for tag, Hd, I, Hkv, r, Bz, T in (("small", 64, 128, 32, 8, 2, 5), ("mid", 256, 512, 64, 16, 2, 64)):
        mk = lambda o, i: (rnd(o, i, scale=0.05), rnd(r, i, dtype=torch.float32, scale=0.05),
                           rnd(o, r, dtype=torch.float32, scale=0.05), 2.0)

        @case(f"lora_mlp_{tag}")
        def _():
            X = rnd(Bz, T, Hd, scale=0.5)
            gate, up, down = mk(I, Hd), mk(I, Hd), mk(Hd, I)
            leaves = [t.to(dev).clone().requires_grad_(True) for t in
                      (X, gate[1], gate[2], up[1], up[2], down[1], down[2])]
            Xg, gA, gB, uA, uB, dA, dB = leaves
            out = fl.LoRA_MLP.apply(Xg * 1.0, gate[0].to(dev), None, gA, gB, gate[3], up[0].to(dev), None, uA, uB,
                                    up[3], down[0].to(dev), None, dA, dB, down[3], sw.swiglu_fg_kernel,
                                    sw.swiglu_DWf_DW_dfg_kernel, False)
            dY = rnd(Bz, T, Hd)
            out.backward(dY.to(dev))
            return dict(X=X, gate=gate, up=up, down=down, dY=dY, out=cpu(out), grads=[cpu(t.grad) for t in leaves])

        @case(f"lora_qkv_{tag}")
        def _():
            X = rnd(Bz, T, Hd, scale=0.5)
            q, k, v = mk(Hd, Hd), mk(Hkv, Hd), mk(Hkv, Hd)
            leaves = [t.to(dev).clone().requires_grad_(True) for t in (X, q[1], q[2], k[1], k[2], v[1], v[2])]
            Xg = leaves[0]
            p = leaves[1:]
            Qo, Ko, Vo = fl.LoRA_QKV.apply(Xg * 1.0, q[0].to(dev), None, p[0], p[1], q[3], k[0].to(dev), None, p[2],
                                           p[3], k[3], v[0].to(dev), None, p[4], p[5], v[3], False)
            dQ, dK, dV = rnd(Bz, T, Hd), rnd(Bz, T, Hkv), rnd(Bz, T, Hkv)
            torch.autograd.backward([Qo, Ko, Vo], [dQ.to(dev), dK.to(dev), dV.to(dev)])
            return dict(X=X, q=q, k=k, v=v, dQ=dQ, dK=dK, dV=dV, Q=cpu(Qo), K=cpu(Ko), V=cpu(Vo),
                        grads=[cpu(t.grad) for t in leaves])

        @case(f"lora_w_{tag}")
        def _():
            X = rnd(Bz, T, Hd, scale=0.5)
            o = mk(Hd, Hd)
            leaves = [t.to(dev).clone().requires_grad_(True) for t in (X, o[1], o[2])]
            O = fl.LoRA_W.apply(leaves[0] * 1.0, o[0].to(dev), None, leaves[1], leaves[2], o[3])
            dY = rnd(Bz, T, Hd)
            O.backward(dY.to(dev))
            return dict(X=X, o=o, dY=dY, out=cpu(O), grads=[cpu(t.grad) for t in leaves])

    # ------------------------------------------------------------------ harness check: native fp16 vs the interpreter fixture
    check = {}
    try:
        old = torch.load(os.path.join(HERE, "..", "tests", "golden", "ref_triton.pt"), weights_only=False)
        c = old["rms_f16_gemma0"]
        Y = rms.Fast_RMS_Layernorm.apply(c["X"].to(dev), c["W"].to(dev), c["eps"], False)
        check["rms_f16_maxdiff"] = float((cpu(Y).float() - c["Y"].float()).abs().max())
        c = old["glu_f16"]
        h = sw.swiglu_fg_kernel(c["e"].to(dev), c["g"].to(dev))
        check["swiglu_f16_maxdiff"] = float((cpu(h).float() - c["swiglu_h"].float()).abs().max())
        c = old["ce_plain_f16"]
        loss = ce.fast_cross_entropy_loss(c["logits"].to(dev), c["labels"].to(dev))
        check["ce_f16_lossdiff"] = float((cpu(loss).float() - c["loss"].float()).abs())
        c = old["rope_f16"]
        Qo, Ko = rope.fast_rope_embedding(c["Q"].to(dev).clone(), c["K"].to(dev).clone(), c["cos"].to(dev),
                                          c["sin"].to(dev), c["idx"].to(dev))
        check["rope_f16_maxdiff"] = float((cpu(Qo).float() - c["Q_idx"].float()).abs().max())
    except Exception as e:
        check["error"] = repr(e)
    G["_meta"] = dict(torch=torch.__version__, triton=triton.__version__, device=torch.cuda.get_device_name(0),
                      arch=getattr(torch.cuda.get_device_properties(0), "gcnArchName", "?"),
                      device_type=getattr(ku, "DEVICE_TYPE", None), reference_root=ref, errors=errors,
                      native_vs_interpreter_fp16=check,
                      note="outputs of the reference's own Triton kernels / autograd Functions, bf16, run natively")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    torch.save(G, a.out)
    print("wrote", os.path.abspath(a.out), os.path.getsize(a.out), "bytes;", len(G) - 1, "cases;", len(errors), "errors")
    print("native-vs-interpreter fp16:", check)
    for k, v in errors.items():
        print("----", k)
        print(v)


if __name__ == "__main__":
    main()
