"""Cross-check of the NF4 restatement against bitsandbytes itself -- TEST INFRASTRUCTURE ONLY.

bitsandbytes is not installed in the build image and there is no network, so the NF4 byte format is
pinned only to its published description (oracle/nf4_ref.c, oracle/ref_ops.py). This script is the cross-check that
closes the gap THE MOMENT the package is importable (or a real `*-bnb-4bit` safetensors file is at hand):

    python oracle/check_against_bitsandbytes.py                       # needs `import bitsandbytes` + a GPU it supports
    python oracle/check_against_bitsandbytes.py --safetensors DIR     # a local `*-bnb-4bit` checkpoint directory

It compares, bit for bit:
  1. bitsandbytes.functional.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=True)
       -> packed bytes, uint8 absmax, nested absmax2 / code2 / offset      vs  oracle.ref_ops.nf4_quantize_np (first level)
  2. bitsandbytes.functional.dequantize_4bit(packed, state)               vs  oracle.ref_ops.nf4_dequantize_state
  3. (with --safetensors) every `*.weight` + `.absmax / .quant_map / .nested_absmax / .nested_quant_map /
     .quant_state.bitsandbytes__nf4` group of the checkpoint decoded by unsloth_amd.checkpoint's reader and by the oracle,
     and -- when bitsandbytes is importable too -- by bitsandbytes.
Exit code 0 = every comparison bit-identical, 1 = a mismatch (printed), 2 = nothing could be checked (skipped).
Reference call sites this format serves: unsloth/kernels/utils.py:650-675 (cdequantize_blockwise_*_nf4 through
fast_dequantize), unsloth/models/llama.py:2615-2626 (BitsAndBytesConfig(nf4, double_quant))."""
import argparse
import glob
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_ops as R  # noqa: E402


def _try_bnb():
    try:
        import bitsandbytes as bnb
        from bitsandbytes import functional as BF
        return bnb, BF
    except Exception as e:                     # not installed / no supported device
        print(f"[skip] bitsandbytes is not importable here ({type(e).__name__}: {e})")
        return None, None


def check_functional(BF, device):
    bad = 0
    g = torch.Generator().manual_seed(0)
    for shape in ((64, 64), (1024, 4096), (4096, 14336), (128, 192)):
        for dtype in (torch.bfloat16, torch.float16):
            W = (torch.randn(shape, generator=g) * 0.02).to(dtype)
            packed, state = BF.quantize_4bit(W.to(device), blocksize=64, quant_type="nf4", compress_statistics=True)
            # 1. first-level quantisation of the oracle on the same values
            p_np, absmax = R.nf4_quantize_np(W.float().numpy().reshape(-1), 64)
            same_bytes = np.array_equal(packed.cpu().numpy().reshape(-1), p_np)
            # 2. decode: bitsandbytes vs the oracle, from bitsandbytes' own state
            ours = R.nf4_dequantize_state(packed.cpu(), _cpu(state), dtype)
            theirs = BF.dequantize_4bit(packed, state).cpu()
            same_decode = torch.equal(ours.view(-1), theirs.view(-1))
            print(f"{tuple(shape)} {dtype}: packed bytes {'==' if same_bytes else '!='}  decode {'==' if same_decode else '!='}")
            bad += (not same_bytes) + (not same_decode)
    return bad


def _cpu(state):
    import copy
    s = copy.copy(state)
    s.absmax = s.absmax.cpu()
    s.code = s.code.cpu() if s.code is not None else None
    if getattr(s, "state2", None) is not None:
        s.state2 = copy.copy(s.state2)
        s.state2.absmax = s.state2.absmax.cpu()
        s.state2.code = s.state2.code.cpu()
        s.offset = s.offset.cpu()
    return s


def check_checkpoint(path, BF, device):
    from safetensors import safe_open
    from unsloth_amd import nf4 as N
    bad = n = 0
    for f in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
        with safe_open(f, framework="pt") as sf:
            keys = set(sf.keys())
            for k in sorted(keys):
                if not k.endswith(".weight") or (k + ".quant_state.bitsandbytes__nf4") not in keys:
                    continue
                packed = sf.get_tensor(k)
                comp = {s: sf.get_tensor(k + "." + s) for s in ("absmax", "quant_map", "nested_absmax", "nested_quant_map",
                                                                "quant_state.bitsandbytes__nf4") if (k + "." + s) in keys}
                qs = N.QuantState.from_dict(comp, torch.device("cpu"))         # the product's reader (checkpoint.py:175-258)
                ours = R.nf4_dequantize_state(packed, qs, qs.dtype)
                n += 1
                if BF is not None:
                    from bitsandbytes.functional import QuantState
                    st = QuantState.from_dict({kk: v for kk, v in comp.items()}, device=device)
                    theirs = BF.dequantize_4bit(packed.to(device), st).cpu()
                    ok = torch.equal(ours.view(-1), theirs.view(-1))
                    bad += not ok
                    print(f"{k}: {'==' if ok else '!='} bitsandbytes")
                else:
                    meta = json.loads(bytes(comp["quant_state.bitsandbytes__nf4"].tolist()).decode())
                    ok = tuple(meta["shape"]) == tuple(ours.shape) and torch.isfinite(ours.float()).all()
                    bad += not ok
    print(f"{n} NF4 tensors read from {path}")
    return bad, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--safetensors", default=None)
    a = ap.parse_args()
    bnb, BF = _try_bnb()
    device = "cuda" if torch.cuda.is_available() else "cpu"
    checked = bad = 0
    if BF is not None:
        try:
            bad += check_functional(BF, device)
            checked += 1
        except Exception as e:
            print(f"[skip] bitsandbytes could not quantise on {device}: {type(e).__name__}: {e}")
    if a.safetensors:
        b, n = check_checkpoint(a.safetensors, BF, device)
        bad += b
        checked += n > 0
    if not checked:
        print("nothing checked (parity of the NF4 format stays UNPINNED: see oracle/nf4_ref.c header)")
        return 2
    print("MISMATCH" if bad else "bit-identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
