#!/usr/bin/env python
"""Per-K-tile instruction census of a GEMM main loop: hipBLASLt's kernel beside ours. ISA only -- no GPU needed.

    python tools/isa_census.py [--out profiles/r06_gemm_isa_census.md]

VERDICT r05 #1(a): "take hipBLASLt's `Custom_Cijk_...MT256x256x64_MI16x16x1` code object for these shapes, llvm-objdump it,
commit a per-K-tile instruction census beside ours". What it does:
  * hipBLASLt: unbundles torch's own copy of the library (the one `torch.matmul` dispatches to on the GPU box:
    <torch>/lib/hipblaslt/library/TensileLibrary_BB_BB_HA_Bias_SAV_UA_Type_BB_HPA_Contraction_l_Alik_Bljk_Cijk_Dijk_gfx950.co,
    a compressed clang offload bundle), disassembles the kernel rocprofv3 names on the step's NT shapes
    (profiles/r02_hipblaslt_kernels.txt) and takes its steady-state loop `label_LoopBeginL0 .. label_LoopBeginL1`
    (ONE K tile of 64 per trip; L1 is the same body for the other SIMD parity, see the census notes);
  * ours: compiles csrc/gemm256.hip with the build's flags and --save-temps and takes the innermost loop of every gemm
    kernel instance asked for (two K tiles per trip for the 8-wave kernels, one for gemm_nt256s);
  * classifies every instruction (MFMA / ds_read / LDS-DMA / s_waitcnt / s_barrier / other VALU / other SALU / s_nop /
    s_setprio / branch), normalises per K TILE and per WAVE, and adds what follows from the counts: waves per workgroup, flops
    per MFMA-issue, LDS bytes read per flop, issue slots that are not MFMA per MFMA, and the histogram of "how many
    non-MFMA instructions sit between two consecutive MFMAs" (the r05 mfma_issue_probe says one wave per SIMD keeps the
    pipe's full rate with <= 3).
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LLVM = "/opt/rocm/lib/llvm/bin"
HBL_KERNEL = "Custom_Cijk_Alik_Bljk_BBS_BH_Bias_HA_S_SAV_NTD_SK3_UserArgs_MT256x256x64_MI16x16x1_shortname0_gfx950"
HBL_CO = "TensileLibrary_BB_BB_HA_Bias_SAV_UA_Type_BB_HPA_Contraction_l_Alik_Bljk_Cijk_Dijk_gfx950.co"

CLASSES = ("mfma", "ds_read", "lds_dma", "vmem_other", "waitcnt", "barrier", "valu", "salu", "nop", "setprio", "branch")


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if (op.startswith("buffer_load") or op.startswith("global_load")) and (" lds" in ins or "_lds_" in op):
        return "lds_dma"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_", "ds_write", "ds_store")):
        return "vmem_other"
    if op == "s_waitcnt":
        return "waitcnt"
    if op == "s_barrier":
        return "barrier"
    if op == "s_nop":
        return "nop"
    if op == "s_setprio":
        return "setprio"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "salu"


def lds_bytes(ins):
    op = ins.split()[0]
    m = re.search(r"_b(\d+)", op)
    return int(m.group(1)) // 8 * 64 if m else 0


def census(body, k_tiles, waves, tile_mnk, name, notes=""):
    c = collections.Counter(classify(i) for i in body)
    lds = sum(lds_bytes(i) for i in body if classify(i) == "ds_read")
    gaps, run = collections.Counter(), None
    for i in body:
        if classify(i) == "mfma":
            if run is not None:
                gaps[run] += 1
            run = 0
        elif run is not None:
            run += 1
    flops_tile = 2 * tile_mnk[0] * tile_mnk[1] * tile_mnk[2]
    per = {k: c[k] / k_tiles for k in CLASSES}
    non_mfma = sum(per[k] for k in CLASSES if k != "mfma")
    return dict(name=name, waves=waves, k_tiles=k_tiles, per_wave=per, lds_bytes_wave=lds / k_tiles,
                lds_bytes_wg=lds / k_tiles * waves, lds_b_per_kflop=lds / k_tiles * waves / (flops_tile / 1e3),
                non_mfma_per_mfma=non_mfma / max(per["mfma"], 1), gaps=dict(sorted(gaps.items())), notes=notes,
                mfma_per_simd=per["mfma"] * waves / 4, issue_per_simd=(non_mfma + per["mfma"]) * waves / 4)


# ------------------------------------------------------------------------------------------------ hipBLASLt
def hipblaslt_loop():
    import torch
    co = os.path.join(os.path.dirname(torch.__file__), "lib", "hipblaslt", "library", HBL_CO)
    if not os.path.isfile(co):
        co = os.path.join("/opt/rocm/lib/hipblaslt/library", HBL_CO)
    tmp = tempfile.mkdtemp(prefix="isa_census_")
    elf = os.path.join(tmp, "hbl.elf")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={co}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"], check=True)
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", elf], capture_output=True, text=True, check=True).stdout
    funcs = sorted({int(l.split()[1], 16) for l in syms.split("\n") if " FUNC " in l})
    start = next(int(l.split()[1], 16) for l in syms.split("\n") if " FUNC " in l and l.split()[-1] == HBL_KERNEL)
    stop = next((a for a in funcs if a > start), start + 0x40000)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", f"--start-address={hex(start)}",
                          f"--stop-address={hex(stop)}", elf], capture_output=True, text=True, check=True).stdout
    meta = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    regs = {}
    i = meta.find(HBL_KERNEL)
    if i >= 0:
        blk = meta[max(0, i - 3000):i + 3000]
        for k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "max_flat_workgroup_size"):
            m = re.search(r"\." + k + r":\s+(\d+)", blk)
            if m:
                regs[k] = int(m.group(1))
    lines = dis.split("\n")
    body, on = [], False
    for l in lines:
        if "<label_LoopBeginL0>:" in l:
            on = True
            continue
        if "<label_LoopBeginL1>:" in l:
            break
        if on:
            t = l.strip()
            if not t or t.endswith(":"):
                continue
            t = re.sub(r"\s*//.*$", "", t)
            body.append(t)
    return body, regs, co


# ------------------------------------------------------------------------------------------------ ours
def our_loops(source="gemm256.hip"):
    from unsloth_amd import _build
    tmp = tempfile.mkdtemp(prefix="isa_census_")
    src = os.path.join(_build.CSRC, source)
    subprocess.run([_build._hipcc()] + _build._flags(source) + ["--save-temps", "-c", src, "-o", os.path.join(tmp, "x.o")],
                   cwd=tmp, check=True, capture_output=True)
    asm = next(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s"))
    text = open(asm).read().split("\n")
    kernels, cur, meta = {}, None, {}
    for l in text:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
                cur = None
                continue
            kernels[cur].append(l)
    for l in text:                       # register counts from the kernel descriptors
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", l)
        if m:
            cur = m.group(1)
            meta[cur] = {}
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|accum_offset|group_segment_fixed_size|next_free_sgpr)\s+(\d+)", l)
        if m and cur:
            meta[cur][m.group(1)] = int(m.group(2))
    out = {}
    for k, ls in kernels.items():
        # innermost loops: header blocks + the blocks annotated as belonging to them
        blocks, lab, owner, is_h, acc = [], None, None, False, []
        for l in ls:
            m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
            if m:
                if lab is not None:
                    blocks.append((lab, owner, is_h, acc))
                lab, owner, is_h, acc = m.group(1), None, False, []
                continue
            if "Inner Loop Header" in l:
                is_h = True
            m = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            if m:
                owner = m.group(1)
            t = l.strip()
            if t and not t.startswith((";", ".", "//")):
                acc.append(re.sub(r"\s*;.*$", "", t))
        if lab is not None:
            blocks.append((lab, owner, is_h, acc))
        best = []
        for h in [b[0] for b in blocks if b[2]]:
            short = h.lstrip(".L")
            body = [i for lab_, own, _, ls_ in blocks if lab_ == h or own == short for i in ls_]
            if sum(classify(i) == "mfma" for i in body) > sum(classify(i) == "mfma" for i in best):
                best = body
        # a loop written in inline asm carries no LLVM loop annotation: take the largest ASMSTART..ASMEND block that
        # branches back to a local label of its own
        blk, on = [], False
        for l in ls:
            if "#ASMSTART" in l:
                blk, on = [], True
                continue
            if "#ASMEND" in l:
                on = False
                ins = [re.sub(r"\s*;.*$", "", t.strip()) for t in blk]
                ins = [t for t in ins if t and not t.startswith((";", ".", "//")) and not re.match(r"^\d+:$", t)]
                if any(re.match(r"s_cbranch_scc[01]\s+\d+b", t) for t in ins) and \
                        sum(classify(i) == "mfma" for i in ins) > sum(classify(i) == "mfma" for i in best):
                    best = ins
                continue
            if on:
                blk.append(l)
        if best:
            out[k] = (best, meta.get(k, {}))
    return out


def fmt(rows):
    cols = ["kernel (steady-state K loop)", "waves/WG", "MFMA", "ds_read", "LDS-DMA", "s_waitcnt", "s_barrier", "other VALU", "other SALU",
            "s_nop", "s_setprio", "non-MFMA issues per MFMA", "MFMA per SIMD per K tile", "all issues per SIMD per K tile",
            "LDS read KiB / K tile / WG", "LDS B per kflop"]
    out = ["| " + " | ".join(cols) + " |", "|" + "---|" * len(cols)]
    for r in rows:
        p = r["per_wave"]
        out.append("| " + " | ".join([
            r["name"], str(r["waves"]), f"{p['mfma']:.0f}", f"{p['ds_read']:.0f}", f"{p['lds_dma']:.0f}", f"{p['waitcnt']:.0f}",
            f"{p['barrier']:.0f}", f"{p['valu']:.0f}", f"{p['salu'] + p['branch']:.0f}", f"{p['nop']:.0f}", f"{p['setprio']:.0f}",
            f"{r['non_mfma_per_mfma']:.2f}", f"{r['mfma_per_simd']:.0f}", f"{r['issue_per_simd']:.0f}",
            f"{r['lds_bytes_wg'] / 1024:.0f}", f"{r['lds_b_per_kflop']:.2f}"]) + " |")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--kernels", default="gemm_nt256p_kernelIDF16bLb0ELb1E,gemm_nt256s_kernelIDF16bLi0E",
                    help="comma-separated substrings of our (mangled) kernel names")
    a = ap.parse_args()
    rows, detail = [], []
    body, regs, co = hipblaslt_loop()
    r = census(body, 1, 4, (256, 256, 64), "hipBLASLt `Custom_Cijk_Alik_Bljk_BBS_BH_..._MT256x256x64_MI16x16x1` (NT, bf16)")
    r["regs"] = regs
    rows.append(r)
    ours = our_loops()
    for want in a.kernels.split(","):
        for k, (b, meta) in ours.items():
            if want in k and "DF16b" in k:
                n_mfma = sum(classify(i) == "mfma" for i in b)
                waves = 4 if "256s" in k else 8
                per_tile = 128 if waves == 4 else 64
                kt = max(1, round(n_mfma / per_tile))
                rr = census(b, kt, waves, (256, 256, 64), f"ours `{k[:70]}`")
                rr["regs"] = meta
                rows.append(rr)
    text = ["# K-loop instruction census: hipBLASLt's NT kernel beside ours (per K tile of 64, per wave; `tools/isa_census.py`)", "",
            f"hipBLASLt code object: `{os.path.basename(co)}` (torch's bundled copy), kernel `{HBL_KERNEL}`.", "", fmt(rows), ""]
    for r in rows:
        text.append(f"* **{r['name']}**: registers {r.get('regs')}; per-wave LDS fragment bytes per K tile {r['lds_bytes_wave']:.0f}; "
                    f"non-MFMA instructions between consecutive MFMAs (count: occurrences per trip) {r['gaps']}")
    s = "\n".join(text) + "\n"
    print(s)
    if a.out:
        with open(os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out, "w") as f:
            f.write(s)


if __name__ == "__main__":
    main()
