"""CPU: the C-ABI library builds, loads, and exports every symbol include/unsloth_amd.h declares
(no compute calls without a GPU); the product path fails loudly without a GPU instead of falling
back; nothing under unsloth_amd/ imports the oracle."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "unsloth_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:int|void)\s+(\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_hot_path():
    names = _declared()
    for must in ("uamd_rms_layernorm_fwd", "uamd_rope_embedding_qk", "uamd_swiglu_DWf_DW_dfg",
                 "uamd_cross_entropy_backward", "cdequantize_blockwise_bf16_nf4", "uamd_gemm_nt_nf4", "uamd_lora_xa"):
        assert must in names
    assert len(names) >= 24


def test_library_exports_every_declared_symbol():
    from unsloth_amd import _build, _lib
    _build.build()                                     # hipcc cross-compiles gfx950 without a GPU
    L = _lib.lib()
    for name in _declared():
        assert hasattr(L, name), f"{name} declared in include/unsloth_amd.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in unsloth_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(_declared())
    assert L.uamd_version() >= 1


def test_gemm_group_struct_layout():
    import ctypes
    from unsloth_amd._lib import GemmGroup
    assert ctypes.sizeof(GemmGroup) == 5 * 8 + 4 * 8 + 4 * 4
    assert GemmGroup.N.offset == 72 and GemmGroup.lora_scale.offset == 80


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_silent_cpu_fallback():
    import unsloth_amd.kernels as K
    X = torch.randn(4, 64)
    with pytest.raises(RuntimeError, match="MI355X"):
        K.Fast_RMS_Layernorm.apply(X, torch.ones(64), 1e-5, False)
    with pytest.raises(RuntimeError, match="MI355X"):
        K.swiglu_fg_kernel(X, X)
    with pytest.raises(RuntimeError, match="MI355X"):
        K.lora_linear_forward(X.bfloat16(), [(torch.randn(8, 64).bfloat16(), None, None, None, None)])


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "unsloth_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports oracle/"
