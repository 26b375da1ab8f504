"""Generates tests/golden/ref_triton_layernorm.pt by running the REFERENCE's own LayerNorm Triton kernels
(unsloth/kernels/layernorm.py, imported read-only from /root/reference through the stub harness of
oracle/make_golden_from_reference.py) on the CPU under TRITON_INTERPRET=1. fp32 and fp16 (the interpreter has no bf16).
Run in the build container only: `python oracle/make_golden_layernorm.py`."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden_from_reference as H  # noqa: E402  (sets TRITON_INTERPRET before torch / triton)
import torch  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_triton_layernorm.pt")


def main():
    H.load_reference()
    ln = importlib.import_module("unsloth.kernels.layernorm")
    gen = torch.Generator().manual_seed(3407)
    G = {}
    for name, rows, dim, dtype in (("f32_small", 5, 96, torch.float32), ("f16_small", 7, 128, torch.float16),
                                   ("f32_vit", 3, 1280, torch.float32), ("f16_ragged", 4, 200, torch.float16)):
        X = (torch.randn(rows, dim, generator=gen) * 1.5 + 0.3).to(dtype)
        W = torch.rand(dim, generator=gen).to(dtype)              # uniform like the reference's self-test (:199-200)
        b = torch.rand(dim, generator=gen).to(dtype)
        dY = torch.randn(rows, dim, generator=gen).to(dtype)
        Xr = X.clone().requires_grad_(True)
        Y = ln.Fast_Layernorm.apply(Xr, W, b, 1e-5)
        Y.backward(dY.clone())
        G[name] = dict(X=X, W=W, b=b, dY=dY, eps=1e-5, Y=Y.detach().clone(), dX=Xr.grad.detach().clone())
    G["_meta"] = dict(source="unsloth/kernels/layernorm.py Fast_Layernorm under TRITON_INTERPRET=1", torch=torch.__version__)
    torch.save(G, OUT)
    print("wrote", OUT, {k: tuple(v["X"].shape) for k, v in G.items() if k != "_meta"})


if __name__ == "__main__":
    main()
