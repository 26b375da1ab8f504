"""Build-time invariants of the attention kernels (csrc/attention.hip), checked on the device assembly: attn_bwd_dkdv4_kernel's
accumulators live in AGPRs the compiler does not know about (every instruction touching them is inline asm naming the physical
registers) -- no compiler-generated instruction may use them; and the kernels whose tile loops count their own `vmcnt` must not
reload spilled registers there."""
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
def test_forward_and_dq_kernels_do_not_spill(tmp_path):
    """The tile loops of the forward kernels and of the dQ kernel carry hand-counted `vmcnt` waits that keep LDS-DMA pieces in
    flight across barriers; a scratch reload is a VMEM load whose compiler-inserted wait drains that ring (round 4 found the
    persistent kernel's DMA offsets and its epilogue pointer spilled exactly there). The plain-causal instances must not touch
    scratch at all; the band instances (50 more live registers) may spill a handful outside the hot loop."""
    lines = _device_asm(tmp_path)
    seen = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN.*(attn_fwd_kernel|attn_fwd_ps_kernel|attn_bwd_dq_kernel)\w*):", l)
        if not m:
            continue
        end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm"))
        scratch = sum(1 for t in lines[i:end + 1] if t.strip().startswith("scratch_"))
        band = "Lb1E" in m.group(1)
        seen += 1
        if m.group(2) == "attn_fwd_ps_kernel" or not band:
            assert scratch == 0, (m.group(1), scratch)
        else:
            # (band dQ: 4-5 dwords spilled ACROSS the middle tile loop, stored before it and reloaded after it -- not per tile)
            assert scratch <= 12, (m.group(1), scratch)
    assert seen == 12                 # 3 kernels x {bf16, fp16} x {plain, band}
