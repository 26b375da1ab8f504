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
def test_dkdv4_accumulators_are_asm_owned_and_step_loops_are_whole(tmp_path):
    """attn_bwd_dkdv4_kernel: ALL 256 AGPRs are the dK^T / dV^T accumulators, owned by inline asm (attn_acc256.inc and the
    generated step loops, attn_kd4_loop.inc). No compiler-generated instruction may name an AGPR. The steps over whole tiles run
    in two generated loops per instance (plain and masked): each is ONE asm statement with both ring stages' bodies (2 x 64
    MFMAs), a backward branch, counted waits only and no scratch access; the C++ body of the ragged tiles stays (64 MFMAs per
    stage)."""
    lines = _device_asm(tmp_path)
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*attn_bwd_dkdv4_kernel.*:", l)]
    assert len(kernels) == 2                                    # bf16 and fp16
    for start in kernels:
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        in_asm, block, loops, cxx_mfma = False, [], [], 0
        for l in lines[start:end + 1]:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm, block = True, []
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                if sum(b.startswith("v_mfma") for b in block) == 128:
                    loops.append(block)
                else:
                    cxx_mfma += sum(b.startswith("v_mfma") for b in block)
                continue
            if t.startswith(";") or not t:
                continue
            if in_asm:
                block.append(t)
            else:
                assert not re.search(r"\ba\d+\b|\ba\[\d+:\d+\]|v_accvgpr", t), f"compiler-generated AGPR use: {t}"
        assert len(loops) == 2, len(loops)                      # plain, masked
        assert cxx_mfma == 128, cxx_mfma                        # the C++ body: 64 MFMAs x 2 stages, one asm statement each
        for block in loops:
            assert any(re.match(r"s_cbranch_scc0\s+1b", b) for b in block)
            assert not any(b.startswith("scratch_") for b in block)
            assert sum(b.startswith("global_load_lds") for b in block) == 34          # 17 pieces per stage
            waits = [b for b in block if b.startswith("s_waitcnt")]
            assert all(re.fullmatch(r"s_waitcnt (lgkmcnt|vmcnt)\(\d+\)", w) for w in waits), waits
            off = 0                                             # 8-byte instructions on 8-byte boundaries inside the bodies
            body = block[block.index("1:"):]
            for b in body:
                op = b.split()[0]
                if op.endswith(":"):
                    continue
                if op == ".p2align":
                    off = 0
                    continue
                lit = [x for x in b.replace(",", " ").split()[1:] if re.fullmatch(r"-?\d+|0x[0-9a-fA-F]+", x)]
                wide = op.startswith(("v_mfma", "ds_read", "global_load", "v_fma", "v_cvt_pk", "v_med3", "v_bfi", "v_bfe")) or \
                    op.endswith("_e64") or (op == "s_add_u32" and lit and not -16 <= int(lit[0], 0) <= 64)
                assert not (wide and off % 8), f"misaligned 8-byte instruction in a step body: {b}"
                off += 8 if wide else 4


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
    assert seen == 30                 # {forward, dQ} x {bf16, fp16} x {plain, band} x head-dim classes {64, 96, 128} + persistent forward x 6
                                      # (plain, band with the static deal, band with claimed items)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_gemm256s_registers_are_asm_owned_and_nothing_spills(tmp_path):
    """gemm_nt256s_kernel (csrc/gemm256.hip, K loop generated by tools/gen/gen_gemm256s.py): the accumulators a[0:255] and the
    B-operand fragments v[192:255] belong to the inline asm ACROSS statements -- no compiler-generated instruction may name
    them (the buffer descriptors s[84:91] are asm-owned only inside a statement: the kernel's scalar register use reaches
    s91) -- and no instance touches scratch (a reload is a VMEM load with a vmcnt(0) behind it: a drain of
    the DMA ring). All eight instances: {bf16, fp16} x {NT, NN} x {one workgroup per tile, persistent walk}; each carries the
    steady-state loop (128 MFMAs and a backward branch inside one asm block), 8-byte instructions 8-byte aligned."""
    import sys
    sys.path.insert(0, ROOT)
    from unsloth_amd import _build
    out = tmp_path / "gemm256.s"
    cmd = [HIPCC] + _build._flags("gemm256.hip") + ["--cuda-device-only", "-S",
           os.path.join(ROOT, "unsloth_amd", "csrc", "gemm256.hip"), "-o", str(out)]
    subprocess.run([c for c in cmd if c != "-fPIC"], check=True, capture_output=True)
    lines = out.read_text().split("\n")
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*gemm_nt256s_kernel.*Li0E.*:", l)]
    assert len(kernels) == 8, len(kernels)
    owned = re.compile(r"\ba\d+\b|\ba\[\d+:\d+\]|v_accvgpr|\bv(19[2-9]|2[0-4]\d|25[0-5])\b|\bv\[(19[2-9]|2[0-4]\d|25[0-5]):")
    for start in kernels:
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        in_asm, loops, block = False, 0, []
        for l in lines[start:end + 1]:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm, block = True, []
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                if sum(b.startswith("v_mfma") for b in block) == 128 and any(re.match(r"s_cbranch_scc0\s+1b", b) for b in block):
                    loops += 1
                    off = 0                                   # every 8-byte instruction of the loop on an 8-byte boundary
                    for b in block:
                        op = b.split()[0]
                        if op.endswith(":") or op.startswith("."):
                            continue
                        lit = [x for x in b.replace(",", " ").split()[1:] if re.fullmatch(r"-?\d+|0x[0-9a-fA-F]+", x)]
                        wide = op.startswith(("v_mfma", "ds_read", "buffer_load", "v_xor")) or (
                            op in ("s_add_u32", "s_xor_b32") and lit and not -16 <= int(lit[0], 0) <= 64)
                        assert not (wide and off % 8), f"misaligned 8-byte instruction in the K loop: {b}"
                        off += 8 if wide else 4
                continue
            if not t or t.startswith(";"):
                continue
            if in_asm:
                block.append(t)
                continue
            assert not t.startswith("scratch_"), f"{lines[start][:80]}: scratch access: {t}"
            assert not owned.search(t), f"{lines[start][:80]}: compiler-generated use of an asm-owned register: {t}"
        assert loops >= 2, (lines[start][:80], loops)         # the main loop and the rank block's (persistent: + the hand-over)
