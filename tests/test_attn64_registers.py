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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_compiler_never_touches_the_hidden_accumulators(tmp_path):
    out = tmp_path / "attention.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5", "-ffp-contract=off",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "unsloth_amd", "csrc"), "--cuda-device-only", "-S",
           os.path.join(ROOT, "unsloth_amd", "csrc", "attention.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    lines = out.read_text().split("\n")
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
