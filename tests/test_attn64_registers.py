"""Build-time invariant of attn_fwd64_kernel (csrc/attention.hip): its O accumulators live in AGPRs a0..a127 that the
compiler does not know about (every instruction touching them is inline asm naming the physical registers). The kernel is
only correct if NO compiler-generated instruction uses those registers -- checked on the device assembly."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _device_asm(tmp_path):
    import sys
    sys.path.insert(0, ROOT)
    from unsloth_amd import _build
    out = tmp_path / "attention.s"
    cmd = [HIPCC] + _build._flags("attention.hip") + ["--cuda-device-only", "-S",
           os.path.join(ROOT, "unsloth_amd", "csrc", "attention.hip"), "-o", str(out)]
    subprocess.run([c for c in cmd if c != "-fPIC"], check=True, capture_output=True)
    return out.read_text().split("\n")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_dkdv4_accumulators_are_asm_owned_and_step_bodies_do_not_spill(tmp_path):
    """attn_bwd_dkdv4_kernel: ALL 256 AGPRs are the dK^T / dV^T accumulators, owned by inline asm (attn_acc256.inc). No
    compiler-generated instruction may name an AGPR, and the four step bodies (the basic blocks that carry the 64 MFMAs
    of a step) must not touch scratch memory."""
    lines = _device_asm(tmp_path)
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*attn_bwd_dkdv4_kernel.*:", l)]
    assert len(kernels) == 2                                    # bf16 and fp16
    for start in kernels:
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        in_asm = False
        blocks, cur = [], {"mfma": 0, "scratch": 0}
        for l in lines[start:end + 1]:
            t = l.strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                blocks.append(cur)
                cur = {"mfma": 0, "scratch": 0}
                continue
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if t.startswith(";") or not t:
                continue
            if t.startswith("v_mfma"):
                cur["mfma"] += 1
            if t.startswith("scratch_"):
                cur["scratch"] += 1
            if not in_asm:
                assert not re.search(r"\ba\d+\b|\ba\[\d+:\d+\]|v_accvgpr", t), f"compiler-generated AGPR use: {t}"
        blocks.append(cur)
        bodies = [b for b in blocks if b["mfma"] >= 64]
        assert len(bodies) == 4, [b for b in blocks if b["mfma"]]     # {plain, masked} x {stage 0, stage 1}
        assert all(b["mfma"] == 64 and b["scratch"] == 0 for b in bodies), bodies


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_compiler_never_touches_the_hidden_accumulators(tmp_path):
    lines = _device_asm(tmp_path)
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*attn_fwd64_kernel.*:", l)]
    assert len(kernels) == 2                                    # bf16 and fp16
    for start in kernels:
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        in_asm, managed, scratch = False, set(), 0
        for l in lines[start:end + 1]:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if t.startswith("scratch_"):
                scratch += 1
            if in_asm or t.startswith(";"):
                continue
            for m in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", t):
                managed.update([int(m.group(1))] if m.group(1) is not None else range(int(m.group(2)), int(m.group(3)) + 1))
        assert managed and min(managed) >= 128, f"compiler-managed AGPRs overlap the hidden accumulators: {sorted(managed)[:8]}"
        assert scratch == 0, "attn_fwd64_kernel spills"
