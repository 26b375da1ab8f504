#!/usr/bin/env python
"""Which waits does hipcc put into the loops of a kernel file?  ISA-level check, no GPU needed.

    python tools/isa_loop_waits.py [name-substring] [csrc file ...]        (default: every csrc/*.hip)

Compiles each file for gfx950 with the build's own flags (unsloth_amd/_build.py) and --save-temps, finds every loop of every
kernel (label ... branch back to the label), and prints for the loops that contain global loads / stores / LDS-DMA:
instruction count, loads, stores, the sequence of `s_waitcnt vmcnt(N)` in the body and the body offsets of every full
`vmcnt(0)`.

Why: hipcc's waitcnt insertion tracks compiler-visible global loads per destination register and places the wait at the first
use. Three times in round 3 that first use (or a control-flow join it could not see through) sat inside a hand-pipelined loop,
where the wait then runs every iteration on the one hardware counter that the loop's own prefetch -- register loads issued a
tile ahead, or LDS-DMA in inline asm -- shares:
  * glu_xa_kernel: a conditional second tile inside the loop = a join -> `vmcnt(0)` at the loop header (prefetch AND the previous
    tile's stores drained every iteration; 291 -> 254 us once the body was branch-free),
  * gemm_nt256p_kernel: the accumulate / bias loads of the epilogue on the persistent walk's back-edge -> `vmcnt(0)` in the K
    loop's header (the counted vmcnt(3) of the DMA ring overridden; +1 % with a load-free epilogue instance),
  * attn_fwd_kernel / attn_bwd_dq_kernel: the resident Q / dO fragments of the prologue -> a vmcnt(7)..vmcnt(0) countdown in
    front of the first eight MFMAs of every tile step (measured neutral: other waves cover it).
A loop whose waits are only the counted ones written in the source prints e.g. `vmcnt sequence ['3', '3']`."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def loops_of(asm_path):
    """{kernel: [(header label, instructions, loads, stores, [vmcnt waits], [body offsets of full vmcnt(0)]), ...]} for every
    innermost loop that touches global memory. A loop = the header block (`; =>This Inner Loop Header`) plus every block
    LLVM annotates with `in Loop: Header=<that block>` -- wherever the layout put them (a rotated loop keeps its latch
    BEFORE the header and falls through into it: a "label ... branch back to the label" scan misses those)."""
    lines = open(asm_path).read().split("\n")
    out, kern = {}, None
    blocks = []                                      # (label, header it belongs to or None, is_header, [lines]) of the current kernel

    def flush():
        if kern is None:
            return
        headers = [b[0] for b in blocks if b[2]]
        for h in headers:
            short = h.lstrip(".L")                   # annotations say BB2_3 for .LBB2_3
            body = []
            for lab, owner, is_h, ls in blocks:
                if lab == h or owner == short:
                    body += ls
            body = [b for b in body if b.strip() and not b.strip().startswith(";")]
            nld = sum(("global_load" in b or "buffer_load" in b) for b in body)
            nst = sum(("global_store" in b or "buffer_store" in b) for b in body)
            if not (nld or nst):
                continue
            waits = [re.search(r"vmcnt\((\d+)\)", b).group(1) for b in body if "vmcnt(" in b]
            drains = [k for k, b in enumerate(body) if re.search(r"s_waitcnt\s+vmcnt\(0\)", b)]
            out.setdefault(kern, []).append((h, len(body), nld, nst, waits, drains))

    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            flush()
            kern, blocks = m.group(1), []
            continue
        if ".Lfunc_end" in l:
            flush()
            kern, blocks = None, []
            continue
        if kern is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        if m:
            blocks.append([m.group(1), None, False, []])
            note = m.group(2)
        elif blocks and re.match(r"^\s+;\s+(=>|Parent Loop|Child Loop|in Loop)", l):
            note = l                                  # the annotation of a nested loop's block continues on comment lines
        else:
            note = None
            if blocks:
                blocks[-1][3].append(l)
        if note is not None and blocks:
            owner = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
            if owner:
                blocks[-1][1] = owner.group(1)
            if "Inner Loop Header" in note:            # innermost loops only: the hand-pipelined ones
                blocks[-1][2] = True
    flush()
    return out


def main():
    from unsloth_amd import _build
    args = sys.argv[1:]
    want = args[0] if args and not args[0].endswith(".hip") else ""
    files = [a for a in args if a.endswith(".hip")] or sorted(f for f in os.listdir(_build.CSRC) if f.endswith(".hip"))
    hipcc = _build._hipcc()
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            src = f if os.path.isabs(f) else os.path.join(_build.CSRC, os.path.basename(f))
            r = subprocess.run([hipcc] + _build._flags(os.path.basename(src)) + ["-c", src, "-o", os.path.join(tmp, "x.o"),
                                                                                 "--save-temps=obj"], capture_output=True, text=True, cwd=tmp)
            if r.returncode != 0:
                print(f"{f}: hipcc failed\n{r.stderr[-2000:]}")
                continue
            asm = [os.path.join(tmp, n) for n in os.listdir(tmp) if n.endswith("gfx950.s")]
            for a in asm:
                for kern, loops in loops_of(a).items():
                    dem = subprocess.run(["c++filt", kern], capture_output=True, text=True).stdout.strip() or kern
                    if want and want not in dem and want not in kern:
                        continue
                    print(f"{os.path.basename(src)} | {dem[:140]}")
                    for lab, n, nld, nst, waits, drains in loops:
                        print(f"     loop {lab}: {n} instrs, {nld} loads, {nst} stores; vmcnt sequence {waits[:16]}"
                              + (f"; FULL vmcnt(0) at body offsets {drains[:6]}" if drains else ""))
                os.remove(a)


if __name__ == "__main__":
    main()
