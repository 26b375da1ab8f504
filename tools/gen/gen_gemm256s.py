"""Generates unsloth_amd/csrc/gemm256s_loop.inc: the hand-scheduled K-tile bodies of gemm_nt256s_kernel (gemm256.hip).

    python tools/gen/gen_gemm256s.py > unsloth_amd/csrc/gemm256s_loop.inc

One wave per SIMD (4 waves x 128 x 128 wave tiles, 256 accumulator AGPRs): nothing else fills the matrix pipe, so the wave
has to pipeline itself -- instruction ORDER is the design, and it is written here instead of being left to hipcc's
scheduler. The order follows the steady-state loop of the vendor kernel for these shapes, read from its disassembly
(`tools/isa_census.py`, profiles/r06_gemm_isa_census.md): per K tile of 64 and per wave 128 MFMAs, 32 ds_read_b128, 16 LDS-DMA
pieces, 3 barriers, and never more than two single-issue instructions between two consecutive MFMAs.

Tile t is multiplied out of LDS stage s = t & 1 in two k-halves of 32 (fragment register sets 0 / 1: 8 A-operand + 8
B-operand fragments each). While the 64 MFMAs of k-half 0 run, the fragments of k-half 1 are read (A first, then B); as soon
as EVERY wave has finished reading an operand's half of the stage (barrier 1: A, barrier 2: B) that half receives tile t + 2
by LDS-DMA. While the 64 MFMAs of k-half 1 run, the rest of tile t + 2's pieces are issued, `s_waitcnt vmcnt(13)` retires
tile t + 1 (13 = the pieces of tile t + 2 issued so far: the counter never drains), barrier 3 publishes it, and the
k-half-0 fragments of tile t + 1 are read from the other stage.

Three bodies: DMA (steady state; also the loop), NODMA (tile nk - 2: nothing left to fetch, vmcnt(0)), LAST (tile nk - 1:
nothing to fetch, nothing to read). Operands are named; the operand lists are macros too, so the kernel binds them once.
B-operand = the matrix whose rows become output COLUMNS (the weight), read as MFMA source A (the lane then holds four
consecutive n of one m: 8-byte stores); A-operand = the activations, MFMA source B. acc[x * 8 + y] = n-tile x, m-tile y.
"""
import sys

NT_LDS = [c * 4096 for c in range(8)]         # piece c of a wave: sub-tile c * 4 + w, 1 KiB each


def mfma(h, x, y):
    q = x * 8 + y
    return f"v_mfma_f32_16x16x32_\" TS \" %[acc{q}], %[x{h}_{x}], %[y{h}_{y}], %[acc{q}]"


def rd_y(h, y):
    return f"ds_read_b128 %[y{h}_{y}], %[rdA{h}] offset:{y * 2048}"


def rd_x(h, x):
    return f"ds_read_b128 %[x{h}_{x}], %[rdB{h}] offset:{x * 2048}"


def dma(op, c):
    """piece c of operand op ('A' / 'B'): the load, then (after the next MFMA) the soffset advance; M0 for the NEXT piece."""
    return f"buffer_load_dwordx4 %[voff{op}], %[srd{op}], %[so{op}{c}] offen lds"


def m0_for(op, c, lds=NT_LDS):
    if c == 0:
        return f"s_mov_b32 m0, %[m0{op}]"
    return f"s_add_u32 m0, %[m0{op}], {lds[c]}"


def adv(op, c):
    return f"s_add_u32 %[so{op}{c}], %[so{op}{c}], %[step{op}]"


def knock(instrs, drop):
    """timing experiments only (results are garbage): drop 'dma' (pieces + their SALU), 'read' (fragment reads), 'sync'
    (waits + barriers)"""
    out = []
    for i in instrs:
        op = i.split()[0]
        if "dma" in drop and (op.startswith("buffer_load") or (op.startswith("s_") and ("m0" in i or "%[so" in i))):
            continue
        if "read" in drop and op.startswith("ds_read"):
            continue
        if "sync" in drop and op in ("s_waitcnt", "s_barrier"):
            continue
        out.append(i)
    return out


def body(kind, shift=0):
    """kind: 'dma' | 'nodma' | 'last'. Returns the instruction list of one K tile. `shift` moves the LDS-DMA / ds_read
    positions of the middle section by one MFMA (the vendor loop has two bodies selected by SIMD parity, so that the four
    waves of a workgroup do not present their LDS traffic in the same cycle)."""
    after = {i: [] for i in range(-1, 128)}          # instructions issued AFTER MFMA i (-1: before the first)
    before = {i: [] for i in range(128)}             # instructions that must sit directly BEFORE MFMA i
    d = kind == "dma"
    nxt = kind != "last"
    # ---- phase 1: k-half 0 MFMAs 0..63; read A(h=1) behind MFMAs 0,2,..,14
    for y in range(8):
        after[2 * y].append(rd_y(1, y))
    if d:
        after[15].append(m0_for("A", 0))
    # barrier 1: every wave has read the A half of this stage (k-half 0 at the end of the previous tile, k-half 1 now)
    before[21].append("s_waitcnt lgkmcnt(0)")
    after[21].append("s_barrier")
    # A pieces 0..4 interleaved with B(h=1) fragment reads 0..4, then reads 5..7
    s = shift
    pos = 22
    for c in range(5):
        if s == 0:                       # DMA, M0, read
            pd, pr = pos, pos + 2
        else:                            # read, DMA, M0 (the other SIMD parity's order)
            pd, pr = pos + 2, pos + 1
        if d:
            after[pd].append(dma("A", c))
            after[pd + 1].append(m0_for("A", c + 1))
            after[pd + 1].append(adv("A", c))
        after[pr].append(rd_x(1, c))
        pos += 3
    for c, p in ((5, 38), (6, 40), (7, 42)):
        after[p].append(rd_x(1, c))
    # barrier 2: every wave has read the B half of this stage
    before[51].append("s_waitcnt lgkmcnt(0)")
    after[51].append("s_barrier")
    if d:
        for c, p in ((5, 52), (6, 55), (7, 58)):
            after[p + s].append(dma("A", c))
            after[p + 1 + s].append(m0_for("A", c + 1) if c < 7 else m0_for("B", 0))
            after[p + 1 + s].append(adv("A", c))
        after[61 + s].append(dma("B", 0))
        after[62 + s].append(m0_for("B", 1))
        after[62 + s].append(adv("B", 0))
        # ---- phase 2: k-half 1 MFMAs 64..127
        after[64 + s].append(dma("B", 1))
        after[65 + s].append(m0_for("B", 2))
        after[65 + s].append(adv("B", 1))
    if nxt:
        # the fragment read pointers move to the other stage (tile t + 1); stage bit = 0x10000
        after[83].append("v_xor_b32 %[rdA0], 0x10000, %[rdA0]")
        after[83].append("v_xor_b32 %[rdB0], 0x10000, %[rdB0]")
        after[84 - s].append("v_xor_b32 %[rdA1], 0x10000, %[rdA1]")
        after[84 - s].append("v_xor_b32 %[rdB1], 0x10000, %[rdB1]")
    if d:
        for c, p in ((2, 85), (3, 87), (4, 89)):
            after[p - s].append(dma("B", c))
            after[p + 1 - s].append(m0_for("B", c + 1))
            after[p + 1 - s].append(adv("B", c))
    if nxt:
        # tile t + 1 has landed (13 newer pieces in flight) and every wave knows it
        before[92].append("s_waitcnt vmcnt(13)" if d else "s_waitcnt vmcnt(0)")
        after[92].append("s_barrier")
        ypos = [93, 94, 95, 97, 98, 102, 103, 104]
        for y, p in enumerate(ypos):
            after[p].append(rd_y(0, y))
        xpos = [105, 106, 109, 111, 114, 116, 119, 122]
        for x, p in enumerate(xpos):
            after[p].append(rd_x(0, x))
    if d:
        after[96 - s].append(dma("B", 5))
        after[97 - s].append(m0_for("B", 6))
        after[97 - s].append(adv("B", 5))
        after[100 - s].append(dma("B", 6))
        after[101 - s].append(m0_for("B", 7))
        after[101 - s].append(adv("B", 6))
        after[124 - s].append(dma("B", 7))
        after[125].append(adv("B", 7))
        # the DMA destination moves to the other stage for the next tile
        after[126].append("s_xor_b32 %[m0A], %[m0A], 0x10000")
        after[126].append("s_xor_b32 %[m0B], %[m0B], 0x10000")
    if nxt:
        before[127].append("s_waitcnt lgkmcnt(0)")
    out = list(after[-1])
    i = 0
    for h in range(2):
        for x in range(8):
            for y in range(8):
                out += before[i]
                out.append(mfma(h, x, y))
                out += after[i]
                i += 1
    return out


def lit(ins):
    return '    "' + ins + '\\n\\t"'


def emit_macro(name, instrs, tail=()):
    print(f"#define {name}(TS) \\")
    lines = [lit(i) for i in instrs] + [lit(i) for i in tail]
    print(" \\\n".join(lines))
    print()


def main():
    print("// GENERATED by tools/gen/gen_gemm256s.py -- do not edit (the schedule is documented there).")
    print("// clang-format off")
    for s in (0, 1):
        emit_macro(f"G256S_LOOP{s}", ["1:"] + body("dma", s),
                   ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b"])
    tail = ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b"]
    # both bodies in ONE statement, selected by the SIMD id's low bit (HW_ID[4]) -- a C++ if / else around two asm statements
    # makes every "+s" operand a PHI, which hipcc then refuses to keep in SGPRs. %[cnt] doubles as the scratch register.
    tail2 = ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 2b"]
    par = ["s_getreg_b32 %[tmp], hwreg(HW_REG_HW_ID, 4, 1)", "s_cmp_eq_u32 %[tmp], 0", "s_cbranch_scc0 2f"] + \
        ["1:"] + body("dma", 0) + tail + ["s_branch 3f", "2:"] + body("dma", 1) + tail2 + ["3:"]
    emit_macro("G256S_LOOP_PAR", par)
    for name, drop in (("KND", {"dma"}), ("KNR", {"read"}), ("KMF", {"dma", "read"}), ("KMO", {"dma", "read", "sync"})):
        emit_macro(f"G256S_LOOP_{name}", ["1:"] + knock(body("dma", 0), drop), tail)
    # experiments (gemm_s4_knock.py): no soffset advance (garbage results); loop head aligned to 64 bytes; shifted by 4 bytes
    emit_macro("G256S_LOOP_XSO", ["1:"] + [i for i in body("dma", 0) if not ("%[so" in i and i.startswith("s_add"))], tail)
    emit_macro("G256S_LOOP_AL64", [".p2align 6", "1:"] + body("dma", 0), tail)
    emit_macro("G256S_LOOP_SH4", [".p2align 6", "s_nop 0", "1:"] + body("dma", 0), tail)
    emit_macro("G256S_NODMA", body("nodma"))
    emit_macro("G256S_LAST", body("last"))
    # a whole tile's 16 pieces back to back (prologue), soffsets advanced, DMA destination flipped
    pro = []
    for op in "AB":
        for c in range(8):
            pro.append(m0_for(op, c))
            pro.append("s_nop 0")
            pro.append(dma(op, c))
            pro.append(adv(op, c))
    pro += ["s_xor_b32 %[m0A], %[m0A], 0x10000", "s_xor_b32 %[m0B], %[m0B], 0x10000"]
    emit_macro("G256S_ISSUE_TILE", pro)
    rd = [rd_y(0, y) for y in range(8)] + [rd_x(0, x) for x in range(8)] + ["s_waitcnt lgkmcnt(0)"]
    emit_macro("G256S_READ0", rd)
    # operand lists
    acc = ", ".join(f'[acc{q}] "+a"(acc[{q}])' for q in range(64))
    fr = ", ".join(f'[{n}{h}_{i}] "+v"({n}f[{h}][{i}])' for n in "yx" for h in range(2) for i in range(8))
    rdp = ", ".join(f'[rd{o}{h}] "+v"(rd{o}[{h}])' for o in "AB" for h in range(2))
    so = ", ".join(f'[so{o}{c}] "+s"(so{o}[{c}])' for o in "AB" for c in range(8))
    print("#define G256S_OUT_ACC " + acc)
    print("#define G256S_OUT_FRAGS " + fr)
    print("#define G256S_OUT_RD " + rdp)
    print("#define G256S_OUT_SO " + so)
    print('#define G256S_OUT_M0 [m0A] "+s"(m0A), [m0B] "+s"(m0B)')
    print('#define G256S_IN_DMA [voffA] "v"(voffA), [voffB] "v"(voffB), [srdA] "s"(srdA), [srdB] "s"(srdB), '
          '[stepA] "s"(stepA), [stepB] "s"(stepB)')
    print("// clang-format on")


if __name__ == "__main__":
    main()
