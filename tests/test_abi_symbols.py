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
    names = re.findall(r"^\s*(?:int|void|int64_t)\s+(\w+)\s*\(", src, flags=re.M)
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


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every struct of include/unsloth_amd.h as gcc sees them == the ctypes / numpy mirrors."""
    import ctypes
    import subprocess
    import numpy as np
    from unsloth_amd._lib import GemmGroup, GemvGroup, LoraTnProblem
    from unsloth_amd.kernels.utils import _PreparedFactors
    fields = {"uamd_gemm_group": [f[0] for f in GemmGroup._fields_],
              "uamd_lora_tn_problem": [f[0] for f in LoraTnProblem._fields_],
              "uamd_gemv_group": [f[0] for f in GemvGroup._fields_],
              "uamd_lora_prep_desc": ["src", "dst_rowmajor", "dst_transposed", "rows", "cols", "dst_pad", "pad_ld",
                                      "pad_scale", "pad_transposed"]}
    src = ["#include <stdio.h>", "#include <stddef.h>", '#include "unsloth_amd.h"', "int main(void) {"]
    for st, fs in fields.items():
        src.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            src.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    src += ["return 0; }"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["uamd_gemm_group"]) == ctypes.sizeof(GemmGroup)
    for f in fields["uamd_gemm_group"]:
        assert int(got[f"uamd_gemm_group.{f}"]) == getattr(GemmGroup, f).offset, f
    assert int(got["uamd_gemv_group"]) == ctypes.sizeof(GemvGroup)
    for f in fields["uamd_gemv_group"]:
        assert int(got[f"uamd_gemv_group.{f}"]) == getattr(GemvGroup, f).offset, f
    assert int(got["uamd_lora_tn_problem"]) == ctypes.sizeof(LoraTnProblem)
    for f in fields["uamd_lora_tn_problem"]:
        assert int(got[f"uamd_lora_tn_problem.{f}"]) == getattr(LoraTnProblem, f).offset, f
    dt = np.dtype(_PreparedFactors._DESC)
    assert int(got["uamd_lora_prep_desc"]) == dt.itemsize
    for cname, (npname, _) in zip(fields["uamd_lora_prep_desc"], _PreparedFactors._DESC):
        assert int(got[f"uamd_lora_prep_desc.{cname}"]) == dt.fields[npname][1], cname


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


def test_flat_adamw_has_no_cpu_fallback():
    """optim.FlatAdamW on host parameters: construction works (it is bookkeeping), the step refuses to run."""
    import pytest
    import torch
    from unsloth_amd.optim import FlatAdamW
    from unsloth_amd.trainer import make_optimizer
    m = torch.nn.Linear(8, 4, bias=False)
    assert isinstance(make_optimizer(m), torch.optim.AdamW) and not isinstance(make_optimizer(m), FlatAdamW)
    opt = FlatAdamW(m)
    m.weight.grad.add_(1.0)
    with pytest.raises(Exception):
        opt.step()


def test_no_cxx_mangled_or_undeclared_exports():
    """The dynamic symbol table of the C-ABI library holds C names only: every defined `uamd_*` / `cdequantize_*` export
    is declared in include/unsloth_amd.h, and nothing C++-mangled leaks out (`_Z15uamd_tuning_geti`)."""
    import re
    import subprocess
    from unsloth_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "unsloth_amd.h")).read()
    declared = set(re.findall(r"\b((?:uamd|cdequantize)_\w+)\s*\(", header))
    mangled = [n for n in names if n.startswith("_Z")]
    assert not mangled, mangled
    stray = [n for n in names if (n.startswith("uamd_") or n.startswith("cdequantize_")) and n not in declared]
    assert not stray, f"exported but not declared in the header: {stray}"


def test_generated_gemm_loop_is_in_sync_with_its_generator():
    """unsloth_amd/csrc/gemm256s_loop.inc is the committed output of tools/gen/gen_gemm256s.py (schedule 'vendor'): the K loop of
    gemm_nt256s_kernel is edited in the generator, never in the include."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("G256S_SCHED", None)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "gen_gemm256s.py")], capture_output=True, text=True,
                         check=True, env=env).stdout
    with open(os.path.join(root, "unsloth_amd", "csrc", "gemm256s_loop.inc")) as f:
        assert f.read() == out


def test_generated_attention_step_loop_is_in_sync_with_its_generator():
    """unsloth_amd/csrc/attn_kd4_loop.inc is the committed output of tools/gen/gen_attn_kd4.py at its default settings: the step
    loops of attn_bwd_dkdv4_kernel are edited in the generator (which also checks the hazards no assembler checks for inline
    asm), never in the include."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("KD4_")}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "gen_attn_kd4.py")], capture_output=True, text=True,
                         check=True, env=env).stdout
    with open(os.path.join(root, "unsloth_amd", "csrc", "attn_kd4_loop.inc")) as f:
        assert f.read() == out
