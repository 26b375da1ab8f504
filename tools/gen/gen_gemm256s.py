"""Generates unsloth_amd/csrc/gemm256s_loop.inc: the hand-scheduled K-tile bodies of gemm_nt256s_kernel (gemm256.hip).

    python tools/gen/gen_gemm256s.py > unsloth_amd/csrc/gemm256s_loop.inc

One wave per SIMD (4 waves x 128 x 128 wave tiles, 256 accumulator AGPRs): nothing else fills the matrix pipe, so the wave
has to pipeline itself -- instruction ORDER is the design, and it is written here instead of being left to hipcc's
scheduler. The skeleton follows the steady-state loop of the vendor kernel for these shapes, read from its disassembly
(`tools/isa_census.py`, profiles/r06_gemm_isa_census.md): per K tile of 64 and per wave 128 MFMAs, 32 ds_read_b128, 16 LDS-DMA
pieces, 3 barriers, and never more than two single-issue instructions between two consecutive MFMAs.

Tile t is multiplied out of LDS stage s = t & 1 in two k-halves of 32 (fragment register sets 0 / 1: 8 A-operand + 8
B-operand fragments each). While the 64 MFMAs of k-half 0 run, the fragments of k-half 1 are read (A first, then B); as soon
as EVERY wave has finished reading an operand's half of the stage (barrier 1: A, barrier 2: B) that half receives tile t + 2
by LDS-DMA. During the 64 MFMAs of k-half 1, `s_waitcnt vmcnt(16)` retires tile t + 1 (16 = the pieces of tile t + 2, all in
flight: the counter never drains), barrier 3 publishes it, and the k-half-0 fragments of tile t + 1 are read from the other
stage.

What round 6's measurements changed against the vendor's placement (profiles/r06d_gemm_s4_knockouts_alignment.jsonl):
  * every DMA piece is issued as EARLY as its half of the stage allows (A: MFMAs 21..35, B: 46..60; the vendor spreads B's
    pieces up to MFMA 124): on K = 14336 the loop is latency-exposed -- a fifth of the pieces miss the XCD's L2 every tile
    -- and the last piece had 96 MFMAs (0.7 us) to land, now 160;
  * the K advance is two 64-bit scalar adds on the buffer descriptors' bases instead of sixteen adds on the piece offsets; the
    descriptors sit in s[84:87] / s[88:91] (an asm operand cannot name half of a register tuple) for the duration of ONE
    statement: built at its entry from the 64-bit `cur` operands, the advanced bases handed back at its exit;
  * the loop head is 64-byte aligned and every 4-byte instruction sits next to another one, so that all 8-byte
    instructions (MFMA, ds_read, buffer_load, literal SALU) stay 8-byte aligned: +2.5 % measured, and a 4-byte shift of
    the same stream costs that much again (MI355X_MICROARCH "code-placement sensitivity").

Bodies: LOOP (steady state), NODMA (tile nk - 2: nothing left to fetch, vmcnt(0)), LAST (tile nk - 1: nothing to fetch,
nothing to read). Operands are named; the operand lists are macros too, so the kernel binds them once.
B-operand = the matrix whose rows become output COLUMNS (the weight), read as MFMA source A (the lane then holds four
consecutive n of one m: 8-byte stores); A-operand = the activations, MFMA source B. acc[x * 8 + y] = n-tile x, m-tile y.
"""
import os

NN = False             # set by main(): B-operand given as [K, N] (the dX products): transposing fragment reads, 2-k-row pieces
SRD = {"A": "s[84:87]", "B": "s[88:91]"}
SRD_LO = {"A": ("s84", "s85"), "B": ("s88", "s89")}
M0_BIAS = 64          # %[m0bA] / %[m0bB] = LDS address of the wave's first piece of that operand in the target stage + 64: every
                      # "m0 = base + k" is then an 8-byte literal add (k = 0 would be an inline constant = a 4-byte instruction)
# LDS offset of a wave's piece c relative to its first piece. Row-major operand tiles ([256 rows][64 k], NT): sub-tile c * 4 + w.
# [64 k][256 n] tiles (NN): a piece is two k-rows; sub-tile u = (w & 1) | (c & 1) << 1 | (w >> 1) << 2 | (c >> 1) << 3, so that the
# bank swizzle f(k-row) = (krow & 3) | ((krow >> 3) & 1) << 2 does not depend on c and ONE per-lane offset serves all pieces.
PIECE_LDS_ROWS = [c * 4096 for c in range(8)]
PIECE_LDS_NN = [(c & 1) * 2048 + (c >> 1) * 8192 for c in range(8)]
XREG0 = 192           # the B-operand fragments live in PINNED v[192:255]: fragment (h, x) = v[192 + 32 h + 4 x : +3]. NN needs it (a
                      # 64-bit transposing read fills HALF a fragment, and an asm operand cannot name half of a register tuple);
                      # NT takes the same registers so that the allocator sees 64 fragment registers instead of 128 (with all of
                      # them as operands the persistent walk spilled twelve around every output tile's source switch -- a scratch
                      # reload is a VMEM load with a vmcnt(0) behind it, i.e. a drain of the DMA ring)


def xreg(h, x, half=None):
    r = XREG0 + 32 * h + 4 * x
    if half is None:
        return f"v[{r}:{r + 3}]"
    return f"v[{r + 2 * half}:{r + 2 * half + 1}]"


def areg(q):
    """accumulator quad q = x * 8 + y: PINNED a[4q : 4q + 3] (all 256 AGPRs, named by the asm and clobbered by every statement:
    as operands the allocator permuted them between statements on the persistent walk and repaired that through scratch)"""
    return f"a[{4 * q}:{4 * q + 3}]"


def mfma(h, x, y):
    q = x * 8 + y
    xs = xreg(h, x)
    return f"v_mfma_f32_16x16x32_\" TS \" {areg(q)}, {xs}, %[y{h}_{y}], {areg(q)}"


def rd_y(h, y):
    return f"ds_read_b128 %[y{h}_{y}], %[rdA{h}] offset:{y * 2048}"


def rd_x(h, x):
    """B-operand fragment x of k-half h: one instruction (row-major tile) or two transposing reads (k-rows 0-3 | 4-7 of the
    lane's 8, 4 x 512 bytes apart; the k-half is 32 k-rows = 16 KiB further)"""
    if NN:
        return [f"ds_read_b64_tr_b16 {xreg(h, x, 0)}, %[rdBn{x}] offset:{h * 16384}",
                f"ds_read_b64_tr_b16 {xreg(h, x, 1)}, %[rdBn{x}] offset:{h * 16384 + 2048}"]
    return [f"ds_read_b128 {xreg(h, x)}, %[rdB{h}] offset:{x * 2048}"]


def dma(op, c):
    return f"buffer_load_dwordx4 %[voff{op}], {SRD[op]}, %[so{op}{c}] offen lds"


def m0_for(op, c):
    off = (PIECE_LDS_NN if (NN and op == "B") else PIECE_LDS_ROWS)[c]
    return f"s_add_u32 m0, %[m0b{op}], {off - M0_BIAS}"


def advance():
    out = []
    for op in "AB":
        lo, hi = SRD_LO[op]
        out += [f"s_add_u32 {lo}, {lo}, %[step{op}]", f"s_addc_u32 {hi}, {hi}, 0"]
    return out


def size(ins):
    op = ins.split()[0]
    if op.startswith(("v_mfma", "ds_read", "buffer_load", "v_xor")):     # (v_xor: VOP2 + 32-bit literal)
        return 8
    toks = ins.replace(",", " ").split()[1:]
    lits = [t for t in toks if t.lstrip("-").isdigit() or t.startswith("0x")]
    if op in ("s_add_u32", "s_xor_b32", "s_mov_b32") and lits:
        v = int(lits[0], 0)
        return 4 if -16 <= v <= 64 else 8
    if op.endswith(":") or op.startswith("."):
        return 0
    return 4


# Where things sit (MFMA index an instruction FOLLOWS). Rules every schedule keeps: A pieces after barrier 1, B pieces after
# barrier 2, X1 reads between the barriers, everything of tile t + 2 issued before `vm` counts as in flight at the vmcnt.
SCHEDULES = {
    # the vendor loop's own placement (profiles/r06_gemm_isa_census.md): pieces never closer than 2-3 MFMAs, B's spread late
    "vendor": dict(y1=[0, 2, 4, 6, 8, 10, 12, 14], bar1=21, A=[22, 25, 28, 31, 34, 52, 55, 58], x1=[24, 27, 30, 33, 36, 38, 40, 42],
                   bar2=51, B=[61, 64, 85, 87, 89, 96, 100, 124], xor=83, bar3=92, y0=[93, 94, 95, 97, 98, 102, 103, 104],
                   x0=[105, 106, 109, 111, 114, 116, 119, 122]),
    # every piece as early as its half of the stage allows, one per two MFMAs (measured: 13 % SLOWER on K = 14336 -- a piece
    # issued right behind another one waits for it in the address path)
    "early": dict(y1=[0, 2, 4, 6, 8, 10, 12, 14], bar1=21, A=[21, 23, 25, 27, 29, 31, 33, 35], x1=[22, 24, 26, 28, 30, 32, 34, 36],
                  bar2=46, B=[46, 48, 50, 52, 54, 56, 58, 60], xor=82, bar3=92, y0=[92, 93, 94, 95, 96, 97, 98, 99],
                  x0=[103, 105, 107, 109, 111, 114, 117, 120]),
    # the vendor's placement with the vmcnt / barrier 3 and the next tile's reads moved 10 MFMAs later
    "late3": dict(y1=[0, 2, 4, 6, 8, 10, 12, 14], bar1=21, A=[22, 25, 28, 31, 34, 52, 55, 58], x1=[24, 27, 30, 33, 36, 38, 40, 42],
                  bar2=51, B=[61, 64, 85, 87, 89, 96, 100, 124], xor=83, bar3=102, y0=[102, 103, 104, 105, 106, 107, 108, 109],
                  x0=[110, 112, 114, 116, 118, 120, 122, 123]),
    # one piece per three MFMAs from barrier 1 on, A and B alternating once B's half is free: all 16 issued by MFMA 79
    "even3": dict(y1=[0, 2, 4, 6, 8, 10, 12, 14], bar1=21, A=[22, 28, 34, 40, 46, 52, 58, 64], x1=[24, 26, 30, 32, 36, 38, 42, 44],
                  bar2=48, B=[49, 55, 61, 67, 70, 73, 76, 79], xor=83, bar3=92, y0=[93, 94, 95, 96, 97, 98, 99, 100],
                  x0=[103, 105, 107, 109, 111, 114, 117, 120]),
}
SCHED = os.environ.get("G256S_SCHED", "vendor")


def body(kind, drop=()):
    """kind: 'dma' | 'nodma' | 'last'. One K tile as an instruction list."""
    S = SCHEDULES[SCHED]
    after = {i: [] for i in range(-1, 130)}          # instructions issued AFTER MFMA i
    before = {i: [] for i in range(128)}             # instructions directly BEFORE MFMA i
    d = kind == "dma"
    nxt = kind != "last"
    # ---- phase 1 (k-half 0, MFMAs 0..63): the k-half-1 fragments, A-operand first
    for y in range(8):
        after[S["y1"][y]].append(rd_y(1, y))
    # barrier 1: every wave has read the A half of this stage (k-half 0 at the end of the previous tile, k-half 1 now)
    before[S["bar1"]] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    for x in range(8):
        after[S["x1"][x]] += rd_x(1, x)
    # barrier 2: every wave has read the B half of this stage
    before[S["bar2"]] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    if d:
        for op in "AB":
            for c in range(8):
                p = S[op][c]
                after[p - 1].insert(0, m0_for(op, c))      # M0 one MFMA ahead of its piece
                after[p].insert(0, dma(op, c))
        last = max(S["A"] + S["B"])
        after[last + 1] += advance()                                       # sources: next K tile
        after[last + 2] += ["s_xor_b32 %[m0bA], %[m0bA], 0x10000", "s_xor_b32 %[m0bB], %[m0bB], 0x10000"]   # destination: the other stage
    # ---- phase 2 (k-half 1, MFMAs 64..127)
    if nxt:
        if NN:
            xs = [f"v_xor_b32 %[rdBn{x}], 0x10000, %[rdBn{x}]" for x in range(8)]
            for k in range(4):           # ten pointer flips over five MFMA gaps
                after[S["xor"] - 3 + k] += xs[2 * k:2 * k + 2]
            after[S["xor"] + 1] += ["v_xor_b32 %[rdA0], 0x10000, %[rdA0]", "v_xor_b32 %[rdA1], 0x10000, %[rdA1]"]
        else:
            after[S["xor"]] += ["v_xor_b32 %[rdA0], 0x10000, %[rdA0]", "v_xor_b32 %[rdB0], 0x10000, %[rdB0]"]
            after[S["xor"] + 1] += ["v_xor_b32 %[rdA1], 0x10000, %[rdA1]", "v_xor_b32 %[rdB1], 0x10000, %[rdB1]"]
        # tile t + 1 has landed (only pieces of tile t + 2 are in flight) and every wave knows it
        inflight = sum(1 for p in S["A"] + S["B"] if p < S["bar3"])
        before[S["bar3"]] += [f"s_waitcnt vmcnt({inflight})" if d else "s_waitcnt vmcnt(0)", "s_barrier"]
        for y in range(8):
            after[S["y0"][y]].append(rd_y(0, y))
        for x in range(8):
            after[S["x0"][x]] += rd_x(0, x)
    out = []
    i = 0
    for h in range(2):
        for x in range(8):
            for y in range(8):
                out += before[i]
                if i == 127 and kind == "dma":
                    out += ["s_waitcnt lgkmcnt(0)", "s_sub_u32 %[cnt], %[cnt], 1"]      # SCC = borrow: the trip count is cnt + 1
                elif i == 127 and nxt:
                    out += ["s_waitcnt lgkmcnt(0)", "s_nop 0"]
                out.append(mfma(h, x, y))
                out += after[i]
                i += 1
    out += after[128] + after[129]
    if drop:
        out = knock(out, drop)
    return out


def knock(instrs, drop):
    """timing experiments only (results are garbage): drop 'dma' (pieces + their SALU), 'read' (fragment reads), 'sync'
    (waits + barriers)"""
    out = []
    for i in instrs:
        op = i.split()[0]
        if "dma" in drop and (op.startswith("buffer_load") or (op.startswith("s_") and any(t in i for t in ("m0", "s84", "s85", "s88", "s89")))):
            continue
        if "read" in drop and op.startswith("ds_read"):
            continue
        if "sync" in drop and op in ("s_waitcnt", "s_barrier"):
            continue
        if "vm" in drop and op == "s_waitcnt" and "vmcnt" in i:
            out.append("s_nop 0")
            continue
        if "bar" in drop and op == "s_barrier":
            out.append("s_nop 0")
            continue
        out.append(i)
    return out


def aligned(instrs):
    """s_nop 0 in front of every 8-byte instruction that would start at 4 mod 8 (the layout above needs none in the loop;
    this is the guard that keeps it so when the schedule is edited)"""
    out, off, pads = [], 0, 0
    for ins in instrs:
        n = size(ins)
        if n == 8 and off % 8 == 4:
            out.append("s_nop 0")
            off += 4
            pads += 1
        out.append(ins)
        off += n
    return out, pads


def lit(ins):
    return '    "' + ins + '\\n\\t"'


def emit_macro(name, instrs):
    print(f"#define {name}(TS) \\")
    print(" \\\n".join(lit(i) for i in instrs))
    print()


def srd_in():
    """the buffer descriptors live in s[84:91] only INSIDE a statement: built at its entry from the `cur` operands (32-bit halves: a 64-bit "+s" operand assigned under a branch is a PHI hipcc cannot keep in SGPRs), the
    advanced bases handed back at its exit (held across statements they were fair game for the compiler, whose scalar register
    use of this kernel reaches s91)"""
    out = []
    for op in "AB":
        lo, hi = SRD_LO[op]
        w2, w3 = ("s86", "s87") if op == "A" else ("s90", "s91")
        out += [f"s_mov_b32 {lo}, %[cur{op}lo]", f"s_mov_b32 {hi}, %[cur{op}hi]", f"s_mov_b32 {w2}, -1", f"s_mov_b32 {w3}, 0x20000"]
    return out


def srd_out():
    out = []
    for op in "AB":
        lo, hi = SRD_LO[op]
        out += [f"s_mov_b32 %[cur{op}lo], {lo}", f"s_mov_b32 %[cur{op}hi], {hi}"]
    return out


def loop(instrs):
    b, pads = aligned(instrs)
    return srd_in() + [".p2align 6", "1:"] + b + ["s_cbranch_scc0 1b"] + srd_out(), pads


def emit_family(pfx):
    l, pads = loop(body("dma"))
    print(f"// {'NN' if NN else 'NT'}: schedule '{SCHED}'; steady-state loop: {sum(size(i) for i in l)} bytes, {pads} alignment pads")
    emit_macro(f"{pfx}_LOOP", l)
    if not NN:
        for name, drop in (("KND", {"dma"}), ("KNR", {"read"}), ("KMF", {"dma", "read"}), ("KMO", {"dma", "read", "sync"}),
                           ("KNV", {"vm"}), ("KNB", {"bar"})):
            emit_macro(f"{pfx}_LOOP_{name}", loop(body("dma", drop))[0])
    emit_macro(f"{pfx}_NODMA", [".p2align 3"] + aligned(body("nodma"))[0])
    emit_macro(f"{pfx}_LAST", [".p2align 3"] + aligned(body("last"))[0])
    # a whole tile's 16 pieces back to back (prologue), sources advanced, DMA destination flipped
    pro = []
    for op in "AB":
        for c in range(8):
            pro += [m0_for(op, c), "s_nop 0", dma(op, c)]
    pro += advance() + ["s_xor_b32 %[m0bA], %[m0bA], 0x10000", "s_xor_b32 %[m0bB], %[m0bB], 0x10000"]
    emit_macro(f"{pfx}_ISSUE_TILE", srd_in() + pro + srd_out())
    rd = [rd_y(0, y) for y in range(8)]
    for x in range(8):
        rd += rd_x(0, x)
    emit_macro(f"{pfx}_READ0", rd + ["s_waitcnt lgkmcnt(0)"])
    # operand lists
    fr = ", ".join(f'[y{h}_{i}] "+v"(yf[{h}][{i}])' for h in range(2) for i in range(8))
    rdp = ", ".join(f'[rdA{h}] "+v"(rdA[{h}])' for h in range(2)) + ", " + \
        (", ".join(f'[rdBn{x}] "+v"(rdBn[{x}])' for x in range(8)) if NN else ", ".join(f'[rdB{h}] "+v"(rdB[{h}])' for h in range(2)))
    print(f"#define {pfx}_OUT_FRAGS " + fr)
    print(f"#define {pfx}_OUTW_FRAGS " + fr.replace('"+v"', '"=v"'))          # write-only: the old fragments are dead (READ0 after an epilogue)
    print(f"#define {pfx}_OUT_RD " + rdp)
    xclob = ", " + ", ".join(f'"v{r}"' for r in range(XREG0, XREG0 + 64))
    print(f'#define {pfx}_CLOBBER "memory", "m0", "scc", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", G256S_ACC_CLOBBER' + xclob)
    print(f'#define {pfx}_CLOBBER_C "memory", G256S_ACC_CLOBBER' + xclob)
    print()


def main():
    global NN
    print("// GENERATED by tools/gen/gen_gemm256s.py -- do not edit (the schedule is documented there).")
    print("// clang-format off")
    NN = False
    emit_family("G256S")
    NN = True
    emit_family("G256SN")
    print("#define G256S_ACC_CLOBBER " + ", ".join(f'"a{i}"' for i in range(256)))
    print("// all accumulators = 0 (the trailing s_nop: v_accvgpr_write -> MFMA reading it as SrcC)")
    print("__device__ __forceinline__ void g256s_acc_zero() {")
    print('    asm volatile("' + "\\n\\t".join(f"v_accvgpr_write_b32 a{i}, 0" for i in range(256)) + '\\n\\ts_nop 3" ::: G256S_ACC_CLOBBER);')
    print("}")
    print("// the two accumulator quads (n-tiles 2 XP, 2 XP + 1) of m-tile Y -> 8 floats")
    print("template <int Y, int XP>")
    print("__device__ __forceinline__ void g256s_acc_read(float (&f)[8]) {")
    for y in range(8):
        for xp in range(4):
            kw = "if" if (y, xp) == (0, 0) else "else if"
            regs = [4 * ((2 * xp + t) * 8 + y) + r for t in range(2) for r in range(4)]
            body = "\\n\\t".join(f"v_accvgpr_read_b32 %{j}, a{r}" for j, r in enumerate(regs))
            outs = ", ".join(f'"=v"(f[{j}])' for j in range(8))
            print(f"    {kw} constexpr (Y == {y} && XP == {xp})")
            print(f'        asm volatile("{body}" : {outs} :: G256S_ACC_CLOBBER);')
    print("}")
    so = ", ".join(f'[so{o}{c}] "s"(so{o}[{c}])' for o in "AB" for c in range(8))
    print('#define G256S_OUT_M0 [m0bA] "+s"(m0bA), [m0bB] "+s"(m0bB), [curAlo] "+s"(curAlo), [curAhi] "+s"(curAhi), [curBlo] "+s"(curBlo), [curBhi] "+s"(curBhi)')
    print("#define G256S_IN_SO " + so)
    print('#define G256S_IN_DMA [voffA] "v"(voffA), [voffB] "v"(voffB), [stepA] "s"(stepA), [stepB] "s"(stepB)')
    print("// clang-format on")


if __name__ == "__main__":
    main()
