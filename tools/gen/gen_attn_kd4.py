"""Generates unsloth_amd/csrc/attn_kd4_loop.inc: the hand-scheduled step loop of attn_bwd_dkdv4_kernel (attention.hip).

    python tools/gen/gen_attn_kd4.py > unsloth_amd/csrc/attn_kd4_loop.inc

The kernel runs one wave per SIMD: nothing else fills the matrix pipe or hides an LDS round trip, so the wave pipelines
itself and instruction ORDER is the design. Through round 5 the order of a step (64 MFMAs: S, dP, dV^T, dK^T of 32 query rows
x 64 keys) was hipcc's between `sched_barrier`s: 455 instructions per step, 33 compiler-placed `s_waitcnt`, 4.9 k cycles per step
against 2.0 k of MFMA issue. Here the UNMASKED steps of a pass (everything between the diagonal tiles and the last step) are
ONE asm statement: a loop over steps, two bodies (ring stage 0 / 1), every instruction placed by the list scheduler below and
every wait counted.

One step (unit = the wave's 32 query rows of one head x the block's 64 keys, kh = key half):
    MFMA  0..15  S[kh]  = Q[:, ks] K^T[kh][ks]              A: Q rows (LDS, 1 b128)       B: K^T (resident VGPRs)
    MFMA 16..31  dP[kh] = -Delta + dO[:, ks] V^T[kh][ks]    A: dO rows                    B: V rows (LDS)  C: -Delta preloaded
    MFMA 32..47  dV^T[kh][dt] += dO^T[dt, c] P[kh][c]       A: dO^T (2 transposing reads) B: P packed
    MFMA 48..63  dK^T[kh][dt] += Q^T[dt, c] dS'[kh][c]      A: Q^T                        B: dS' = P (dP - Delta) packed
and beside them (the "fillers" of the 64 MFMA gaps): 48 operand fragments through an 8-slot register ring (64 ds_read), the
stats line (12 reads: -Delta straight into the dP accumulators, LSE2), P = exp2(S c - LSE2) (64 + 16 pack), dS' = P dP' (32 + 16
pack: subtracting Delta is the MFMA's C operand), the next step's 17 LDS-DMA pieces, the source advance, and the first three Q
fragments of the NEXT step (the stream continues across the back edge). Registers: a[0:255] accumulators, v[128:255] below --
owned by the asm (clobbers); everything else arrives as operands.
"""
import os
import sys

TS = '" TS "'                      # dtype suffix, spliced in by the C preprocessor: bf16 | f16
CAP = int(os.environ.get("KD4_CAP", "5"))          # fillers per MFMA gap the scheduler aims for
DIST = int(os.environ.get("KD4_DIST", "2"))        # MFMAs between a fragment read and its first consumer, at least
LOOK = int(os.environ.get("KD4_LOOK", "7"))        # ... and at most (a read issued much earlier only queues in front of later ones)
KEEP = int(os.environ.get("KD4_KEEP", "4"))        # LDS reads a wait leaves in flight, at most
VMGAP = int(os.environ.get("KD4_VMGAP", "44"))      # first MFMA gap the wait for the next step's tile may sit in
DROP = set(filter(None, os.environ.get("KD4_DROP", "").split(",")))    # timing experiments only: dma | valu | lds knocked out
NOPAD = os.environ.get("KD4_NOPAD", "0") == "1"     # experiment: 8-byte instructions left wherever they fall
NR = 8                                               # fragment ring slots

SC = [128, 144]                   # S / P, key half 0 / 1 (16 registers each: row r <-> q = (r & 3) + 8 (r >> 2) + 4 lh)
DP = [160, 176]                   # dP' = dP - Delta
LSE = 208                         # LSE2: two quads in flight (quad g = rows 4 g .. 4 g + 3 in slot g & 1)
MK = [216, 217]                   # masked bodies: bit rho of MK[kh] = the lane's key of half kh attends row q0 + 4 lh + rho
TMP = 218                         # .. 221


def lse_reg(g, e):
    return LSE + 4 * (g & 1) + e


def PB(kh, c):                    # packed P (then dS') as MFMA B operand: rows 16 c .. 16 c + 15 of key half kh
    return 192 + 8 * kh + 4 * c


def FR(slot):
    return 224 + 4 * slot


def vt(r, n=4):
    return f"v[{r}:{r + n - 1}]"


def regs(r, n):
    return {f"v{i}" for i in range(r, r + n)}


class Ins:
    def __init__(self, text, kind, reads=(), writes=(), size=8):
        self.text, self.kind, self.reads, self.writes, self.size = text, kind, set(reads), set(writes), size
        self.alt = None            # (text, size) of the 4-byte encoding, where one exists

    def __repr__(self):
        return self.text


def mfma_list(stage):
    out = []
    for k in range(8):
        for kh in range(2):
            c = "0" if k == 0 else vt(SC[kh], 16)
            out.append(dict(a=("Q", k), text=f"v_mfma_f32_32x32x16_{TS} {vt(SC[kh], 16)}, {{A}}, %[kf{kh}_{k}], {c}",
                            reads=(regs(SC[kh], 16) if k else set()), writes=regs(SC[kh], 16)))
    for ks in range(8):
        for kh in range(2):
            out.append(dict(a=("dO", ks), b=("V", kh, ks), text=f"v_mfma_f32_32x32x16_{TS} {vt(DP[kh], 16)}, {{A}}, {{B}}, {vt(DP[kh], 16)}",
                            reads=regs(DP[kh], 16), writes=regs(DP[kh], 16)))
    for which, base in (("dOT", 0), ("QT", 8)):
        for c in range(2):
            for dt in range(4):
                for kh in range(2):
                    t = 16 * (base + 4 * kh + dt)
                    out.append(dict(a=(which, c, dt), text=f"v_mfma_f32_32x32x16_{TS} a[{t}:{t + 15}], {{A}}, {vt(PB(kh, c))}, a[{t}:{t + 15}]",
                                    reads=regs(PB(kh, c), 4), writes=set()))
    return out


def fragments(stage):
    """the 48 operand fragments of a step in consumption order: (key, [LDS instruction texts with {R} = ring registers], first / last consumer)"""
    so = stage * 16384
    fr = []
    for k in range(8):
        fr.append((("Q", k), [("ds_read_b128 {R4}, %[cq" + str(k) + f"] offset:{so}", 4, 0)], 2 * k, 2 * k + 1))
    for ks in range(8):
        m = 16 + 2 * ks
        fr.append((("dO", ks), [("ds_read_b128 {R4}, %[cq" + str(ks) + f"] offset:{so + 8192}", 4, 0)], m, m + 1))
        fr.append((("V", 0, ks), [("ds_read_b128 {R4}, %[cv" + str(ks) + "]", 4, 0)], m, m))
        fr.append((("V", 1, ks), [("ds_read_b128 {R4}, %[cv" + str(ks) + "] offset:8192", 4, 0)], m + 1, m + 1))
    for which, first, off in (("dOT", 32, 8192), ("QT", 48, 0)):
        for c in range(2):
            for dt in range(4):
                m = first + 8 * c + 2 * dt
                o = so + off + c * 4096
                fr.append(((which, c, dt), [("ds_read_b64_tr_b16 {R2}, %[ct" + str(dt) + f"] offset:{o}", 2, 0),
                                            ("ds_read_b64_tr_b16 {R2}, %[ctb" + str(dt) + f"] offset:{o}", 2, 2)], m, m + 1))
    assert len(fr) == 48
    return fr


class Item:
    def __init__(self, name, ins, earliest, deadline, preds=(), dma=False):
        self.name, self.ins, self.earliest, self.deadline, self.preds, self.dma = name, ins, earliest, deadline, list(preds), dma
        self.gap = None


def build_items(stage, mask=False):
    so = stage * 16384
    items = []
    frs = fragments(stage)
    slot_of = {}
    # ---- operand fragments (ring slot = index mod NR; a slot is free once the last consumer of its previous occupant has issued)
    for j, (key, reads, first, last) in enumerate(frs):
        slot = j % NR
        slot_of[key] = slot
        earliest = max(0 if j < NR else frs[j - NR][3], first - 1 - LOOK)
        if j < 3:
            continue                                   # Q0..Q2: read at the end of the previous step (or by the entry)
        prev = None
        for text, n, sub in reads:
            r = FR(slot) + sub
            t = text.replace("{R4}", vt(r, 4)).replace("{R2}", vt(r, 2))
            it = Item(f"frag{j}", [Ins(t, "lds", writes=regs(r, n))], earliest, first - 1 - DIST, preds=[prev] if prev else [])
            items.append(it)
            prev = it
    # ---- stats: -Delta into both dP accumulators (C operand of their first MFMA), LSE2
    stat = stage * 1024
    for g in range(4):
        for kh in range(2):
            items.append(Item(f"ndelta{kh}{g}", [Ins(f"ds_read_b128 {vt(DP[kh] + 4 * g)}, %[cs] offset:{stat + 128 + 32 * g}", "lds",
                                                     writes=regs(DP[kh] + 4 * g, 4))], 4 + g, 12))
    lse_item = {}
    for g in range(4):
        lse_item[g] = Item(f"lse{g}", [Ins(f"ds_read_b128 {vt(lse_reg(g, 0))}, %[cs] offset:{stat + 32 * g}", "lds",
                                           writes=regs(lse_reg(g, 0), 4))], 8 + g if g < 2 else 17, 15 if g < 2 else 34 + g)
        items.append(lse_item[g])
    # ---- masked bodies: the valid rows of a lane's key are an interval [lo', hi'] of rho = q - q0 - 4 lh (band edges minus the
    #      step's first row, s95): MK[kh] = its bit mask over rho 0..31 (ten instructions per key half, beside the S MFMAs)
    mk_last = {}
    if mask:
        a, b, c = TMP, TMP + 1, TMP + 2
        for kh in range(2):
            seq = [Ins(f"v_subrev_u32_e32 v{a}, s95, %[qlo{kh}]", "valu", writes={f"v{a}"}, size=4),
                   Ins(f"v_subrev_u32_e32 v{b}, s95, %[qhi{kh}]", "valu", writes={f"v{b}"}, size=4),
                   Ins(f"v_med3_i32 v{a}, v{a}, 0, 31", "valu", reads={f"v{a}"}, writes={f"v{a}"}),
                   Ins(f"v_med3_i32 v{c}, v{b}, 0, 31", "valu", reads={f"v{b}"}, writes={f"v{c}"}),
                   Ins(f"v_sub_u32_e32 v{c}, 31, v{c}", "valu", reads={f"v{c}"}, writes={f"v{c}"}, size=4),
                   Ins(f"v_lshlrev_b32_e64 v{a}, v{a}, -1", "valu", reads={f"v{a}"}, writes={f"v{a}"}),
                   Ins(f"v_lshrrev_b32_e64 v{c}, v{c}, -1", "valu", reads={f"v{c}"}, writes={f"v{c}"}),
                   Ins(f"v_and_b32_e32 v{MK[kh]}, v{a}, v{c}", "valu", reads={f"v{a}", f"v{c}"}, writes={f"v{MK[kh]}"}, size=4),
                   Ins(f"v_ashrrev_i32_e32 v{b}, 31, v{b}", "valu", reads={f"v{b}"}, writes={f"v{b}"}, size=4),
                   Ins(f"v_bfi_b32 v{MK[kh]}, v{b}, 0, v{MK[kh]}", "valu", reads={f"v{b}", f"v{MK[kh]}"}, writes={f"v{MK[kh]}"})]
            prev = mk_last.get(0)
            for n, i in enumerate(seq):
                it = Item(f"mk{kh}_{n}", [i], 0, 15, preds=[prev] if prev else [])
                items.append(it)
                prev = it
            mk_last[kh] = prev
        items.append(Item("q0adv", [Ins("s_add_u32 s95, s95, %[rowadv]", "salu", size=4)], 0, 40, preds=[mk_last[1]]))
    # ---- P = exp2(S c - LSE2), quad by quad (kh, g: rows 4 g .. 4 g + 3), packed into PB(kh, g >> 1)[2 (g & 1) ..]
    pq = {}
    for g in range(4):
        for kh in range(2):
            c = g >> 1
            first = 17 + kh                               # S[kh] is complete two MFMAs behind its last one (14 + kh)
            dead = 32 + 8 * c + kh - 2                    # two wait states in front of the first dV MFMA that reads the words
            seq = []
            for e in range(4):
                r = SC[kh] + 4 * g + e
                seq.append(Ins(f"v_fma_f32 v{r}, v{r}, %[sl2], -v{lse_reg(g, e)}", "valu", reads={f"v{r}", f"v{lse_reg(g, e)}"}, writes={f"v{r}"}))
            for e in range(4):
                r = SC[kh] + 4 * g + e
                i = Ins(f"v_exp_f32_e64 v{r}, v{r}", "trans", reads={f"v{r}"}, writes={f"v{r}"})
                i.alt = (f"v_exp_f32_e32 v{r}, v{r}", 4)
                seq.append(i)
            if mask:                                      # P &= -(bit rho(r) of the mask): masked entries 0, hence dS' = 0 too
                for e in range(4):
                    seq.append(Ins(f"v_bfe_i32 v{TMP + e}, v{MK[kh]}, {e + 8 * g}, 1", "valu", reads={f"v{MK[kh]}"}, writes={f"v{TMP + e}"}))
                for e in range(4):
                    r = SC[kh] + 4 * g + e
                    i = Ins(f"v_and_b32_e64 v{r}, v{r}, v{TMP + e}", "valu", reads={f"v{r}", f"v{TMP + e}"}, writes={f"v{r}"})
                    i.alt = (f"v_and_b32_e32 v{r}, v{r}, v{TMP + e}", 4)
                    seq.append(i)
            for h in range(2):
                r = SC[kh] + 4 * g + 2 * h
                w = PB(kh, c) + 2 * (g & 1) + h
                seq.append(Ins(f"v_cvt_pk_{TS}_f32 v{w}, v{r}, v{r + 1}", "valu", reads={f"v{r}", f"v{r + 1}"}, writes={f"v{w}"}))
            prev = None
            for n, i in enumerate(seq):
                pr = [prev] if prev else [lse_item[g]] + ([mk_last[kh]] if mask else []) + ([pq[(1, g - 1)]] if (mask and g) else [])
                if n == 0 and kh == 1 and mask:
                    pr.append(pq[(0, g)])                 # (the four mask temporaries serve one quad at a time)
                if n == 0 and mask:
                    pr.append(mk_last[1])
                it = Item(f"P{kh}{g}_{n}", [i], first, dead, preds=pr)
                items.append(it)
                prev = it
                if n == 3 and g < 2:
                    lse_item[g + 2].preds.append(it)       # the quad's LSE slot is free once both key halves have read it
            pq[(kh, g)] = prev
    # ---- dS' = P dP', packed over the dead P words: PB(kh, c) is last read by dV MFMA 38 + 8 c + kh
    for g in range(4):
        for kh in range(2):
            c = g >> 1
            prev = pq[(kh, g)]
            for e in range(4):
                r, d = SC[kh] + 4 * g + e, DP[kh] + 4 * g + e
                i = Ins(f"v_mul_f32_e64 v{d}, v{r}, v{d}", "valu", reads={f"v{r}", f"v{d}"}, writes={f"v{d}"})
                i.alt = (f"v_mul_f32_e32 v{d}, v{r}, v{d}", 4)
                it = Item(f"dS{kh}{g}_{e}", [i], 33 + kh, 46 + 8 * c + kh, preds=[prev])
                items.append(it)
                prev = it
            for h in range(2):
                d = DP[kh] + 4 * g + 2 * h
                w = PB(kh, c) + 2 * (g & 1) + h
                it = Item(f"dSp{kh}{g}_{h}", [Ins(f"v_cvt_pk_{TS}_f32 v{w}, v{d}, v{d + 1}", "valu", reads={f"v{d}", f"v{d + 1}"}, writes={f"v{w}"})],
                          38 + 8 * c + kh, 46 + 8 * c + kh, preds=[prev])
                items.append(it)
                prev = it
    # ---- the next step's tile into the OTHER stage: 16 pieces of 1 KiB + the stats line; M0 is set right behind the previous piece
    other = (stage ^ 1) * 16384
    pieces = []
    for h in range(4):
        for i in range(4):
            voff = f"%[qo{i}]" if h < 2 else f"%[doo{i}]"
            pieces.append((f"global_load_lds_dwordx4 {voff}, s[{84 + 2 * h}:{85 + 2 * h}]", f"s_add_u32 m0, %[ringu], {other + h * 4096 + i * 1024}"))
    pieces.append(("global_load_lds_dword %[vstat], s[92:93]", f"s_add_u32 m0, %[statu], {(stage ^ 1) * 1024}"))
    prev = None
    for n, (ld, m0) in enumerate(pieces):
        ins = [Ins(ld, "dma", reads={"m0"})]
        if n + 1 < len(pieces):
            ins.append(Ins(pieces[n + 1][1], "salu", writes={"m0"}))
        it = Item(f"dma{n}", ins, 0, 40, preds=[prev] if prev else [], dma=True)
        items.append(it)
        prev = it
    first_m0 = Ins(pieces[0][1], "salu", writes={"m0"})
    # source advance: the tile after the next
    for n, (lo, hi, st) in enumerate(((84, 85, "advq"), (86, 87, "advq"), (88, 89, "advd"), (90, 91, "advd"))):
        it = Item(f"adv{n}", [Ins(f"s_add_u32 s{lo}, s{lo}, %[{st}lo]", "salu", size=4), Ins(f"s_addc_u32 s{hi}, s{hi}, %[{st}hi]", "salu", size=4)],
                  0, 50, preds=[prev])
        items.append(it)
    items.append(Item("advs", [Ins("s_add_u32 s92, s92, %[advs]", "salu", size=4), Ins("s_addc_u32 s93, s93, %[advshi]", "salu", size=4)], 0, 50, preds=[prev]))
    # ---- the next step: its tile has landed (issued 40+ MFMAs ago), its first three Q fragments into ring slots 0..2
    vm = Item("vmcnt", [Ins("s_waitcnt vmcnt(0)", "wait", size=4)], VMGAP, max(56, VMGAP + 2), preds=[prev])    # (behind the last piece)
    items.append(vm)
    nxt = fragments(stage ^ 1)
    for j in range(3):
        text = nxt[j][1][0][0].replace("{R4}", vt(FR(j), 4))
        items.append(Item(f"nextQ{j}", [Ins(text, "lds", writes=regs(FR(j), 4))], frs[40 + j][3], 63, preds=[vm] + ([items[-1]] if j else [])))
    if DROP:
        kinds = {"dma": ("dma", "salu", "wait"), "valu": ("valu", "trans"), "lds": ("lds",)}
        gone = {k for d in DROP for k in kinds[d]}
        keep = [it for it in items if it.ins[0].kind not in gone]
        for it in keep:
            it.preds = [p_ for p_ in it.preds if p_ in keep]
        items = keep
        if "dma" in DROP:
            first_m0 = Ins("s_nop 0", "salu", size=4)
    return items, slot_of, first_m0


def schedule(stage, mask=False):
    items, slot_of, first_m0 = build_items(stage, mask)
    mf = mfma_list(stage)
    changed = True                                       # a predecessor is due no later than what waits for it
    while changed:
        changed = False
        for it in items:
            for p_ in it.preds:
                if p_.deadline > it.deadline:
                    p_.deadline, changed = it.deadline, True
    gaps = [[] for _ in range(64)]
    placed = set()
    pending = list(items)
    for g in range(64):
        n, ndma = 0, 0
        while True:
            ready = [it for it in pending if it.earliest <= g and all(p in placed for p in it.preds)
                     and not (it.dma and ndma >= 1)]
            if not ready:
                break
            urgent = [it for it in ready if it.deadline <= g]
            if n >= CAP and not urgent:
                break
            # the next step's tile goes out first (one piece per gap: it has the longest way), then whatever is due earliest
            it = min(ready, key=lambda x: (not x.dma and not (x.deadline <= g), x.deadline, items.index(x)))
            if n >= CAP and not (it.deadline <= g):
                break
            if it.deadline < g:
                raise SystemExit(f"stage {stage}: {it.name} misses its deadline {it.deadline} at gap {g} (CAP {CAP})")
            gaps[g].append(it)
            it.gap = g
            placed.add(it)
            pending.remove(it)
            n += len(it.ins)
            ndma += it.dma
    if pending:
        raise SystemExit(f"stage {stage}: unplaced {[i.name for i in pending]}")
    # ---- linear stream with operand registers resolved
    stream = [first_m0]
    for m, d in enumerate(mf):
        a = d["a"]
        areg = FR(slot_of[a])
        text = d["text"].replace("{A}", vt(areg))
        reads = set(d["reads"]) | regs(areg, 4)
        if "b" in d:
            breg = FR(slot_of[d["b"]])
            text = text.replace("{B}", vt(breg))
            reads |= regs(breg, 4)
        i = Ins(text, "mfma", reads=reads, writes=d["writes"])
        i.m = m
        stream.append(i)
        for it in gaps[m]:
            stream += it.ins
    return stream, gaps


def add_waits(stream):
    """counted lgkmcnt waits: LDS operations complete in order; at the top of a body the three Q fragments of the step are the
    newest outstanding operations (issued by the previous body or by the entry)"""
    issued = [regs(FR(j), 4) for j in range(3)]          # write sets in issue order
    done = 0                                             # operations [0, done) are known complete
    out = []
    for ins in stream:
        need = -1
        touched = ins.reads | ins.writes                 # (a write over a pending LDS write must wait too)
        for idx in range(done, len(issued)):
            if issued[idx] & touched:
                need = idx
        if need >= 0:
            # a stricter wait is still a correct one: leave only the KEEP youngest reads in flight (issued within the last gap
            # or two) -- everything older has landed long ago, and the consumers of those reads then need no wait of their own
            n = min(len(issued) - 1 - need, KEEP)
            out.append(Ins(f"s_waitcnt lgkmcnt({n})", "wait", size=4))
            done = len(issued) - n
        if ins.kind == "lds":
            issued.append(set(ins.writes))
        out.append(ins)
    return out


def check(stream):
    """the hazards nobody checks for inline asm: MFMA result -> VALU (two MFMAs in between), VALU result -> MFMA operand (two
    instructions), transcendental result -> VALU (one instruction), M0 write -> LDS-DMA (one instruction)"""
    last_w = {}                                          # register -> (position, kind, mfma index at that time)
    nm = 0
    for pos, ins in enumerate(stream):
        for r in ins.reads | (ins.writes if ins.kind in ("valu", "trans", "lds") else set()):
            if r not in last_w:
                continue
            p, kind, m = last_w[r]
            if kind == "mfma" and ins.kind in ("valu", "trans", "lds") and nm - m < 2:
                raise SystemExit(f"MFMA -> {ins.text}: only {nm - m} MFMAs behind the write of {r}")
            if kind in ("valu", "trans") and ins.kind == "mfma" and r in ins.reads and pos - p < 3:
                raise SystemExit(f"VALU -> {ins.text}: {pos - p - 1} wait states behind the write of {r}")
            if kind == "trans" and ins.kind in ("valu", "trans") and r in ins.reads and pos - p < 2:
                raise SystemExit(f"trans -> {ins.text}: no wait state behind the write of {r}")
            if kind == "salu" and r == "m0" and ins.kind == "dma" and pos - p < 2:
                raise SystemExit(f"M0 -> {ins.text}: no wait state")
        if ins.kind == "mfma":
            nm += 1
        for r in ins.writes:
            last_w[r] = (pos, ins.kind, nm)


def aligned(stream):
    """8-byte instructions on 8-byte boundaries (MI355X_MICROARCH 'code-placement sensitivity'): a v_exp / v_mul takes its 4-byte
    encoding where that repairs the parity, otherwise an s_nop 0 pads"""
    out, off, pads = [], 0, 0
    for k, ins in enumerate(stream):
        text, size = ins.text, ins.size
        if off % 8 == 4 and not NOPAD:
            if ins.alt:
                text, size = ins.alt
            elif size == 8:
                out.append("s_nop 0")
                off += 4
                pads += 1
        out.append(text)
        off += size
    return out, off, pads


def size_fix(ins):
    if ins.text.startswith("s_add_u32 m0"):
        v = int(ins.text.split(",")[-1])
        ins.size = 4 if -16 <= v <= 64 else 8
    return ins


def body(stage, mask=False):
    stream, gaps = schedule(stage, mask)
    stream = add_waits([size_fix(i) for i in stream])
    check(stream)
    return stream, gaps


def lit(text):
    return '    "' + text + '\\n\\t"'


def emit_loop(name, mask):
    b0, g0 = body(0, mask)
    b1, g1 = body(1, mask)
    for s_, (b, g) in enumerate(((b0, g0), (b1, g1))):
        fill = [sum(len(it.ins) for it in gg) for gg in g]
        kinds = {}
        for i in b:
            kinds[i.kind] = kinds.get(i.kind, 0) + 1
        print(f"// {name} stage {s_}: {len(b)} instructions {kinds}; fillers per MFMA gap {fill}")
    # entry: sources into s[84:93], wait for the step's tile, its first three Q fragments; then into the body of the entry stage
    entry = ["s_mov_b32 s94, m0"]               # (M0 is a reserved register: handed back at the exit instead of clobbered)
    for n, nm in enumerate(("q0", "q1", "d0", "d1", "st")):
        entry += [f"s_mov_b32 s{84 + 2 * n}, %[{nm}lo]", f"s_mov_b32 s{85 + 2 * n}, %[{nm}hi]"]
    if mask:
        entry.append("s_mov_b32 s95, %[q0s]")
    entry += ["s_waitcnt vmcnt(0)", "s_cmp_eq_u32 %[stage], 0", "s_cbranch_scc0 3f"]
    pre = []
    for s_ in range(2):
        fr = fragments(s_)
        pre.append([fr[j][1][0][0].replace("{R4}", vt(FR(j), 4)) for j in range(3)])
    out = entry + pre[0] + ["s_branch 1f", "3:"] + pre[1] + ["s_branch 2f"]
    t0, _, p0 = aligned(b0)
    t1, _, p1 = aligned(b1)
    out += [".p2align 6", "1:"] + t0 + ["s_sub_u32 %[cnt], %[cnt], 1", "s_cbranch_scc1 4f", ".p2align 3", "2:"] + t1 + \
           ["s_sub_u32 %[cnt], %[cnt], 1", "s_cbranch_scc0 1b", "4:", "s_waitcnt lgkmcnt(0)", "s_mov_b32 m0, s94"]
    print(f"// {name}: alignment pads {p0} / {p1}")
    print(f"#define {name}(TS) \\")
    print(" \\\n".join(lit(t) for t in out))
    print()


def main():
    print("// GENERATED by tools/gen/gen_attn_kd4.py -- do not edit (the schedule is documented there).")
    print("// clang-format off")
    emit_loop("KD4_LOOP", False)
    emit_loop("KD4_LOOP_M", True)
    kf = ", ".join(f'[kf{kh}_{k}] "v"(kf[{kh}][{k}])' for kh in range(2) for k in range(8))
    ad = ", ".join(f'[cq{k}] "v"(cq[{k}]), [cv{k}] "v"(cv[{k}])' for k in range(8)) + ", " + \
        ", ".join(f'[ct{d}] "v"(ct[{d}]), [ctb{d}] "v"(ct2[{d}])' for d in range(4)) + ', [cs] "v"(cs)'
    dm = ", ".join(f'[qo{i}] "v"(qo[{i}]), [doo{i}] "v"(doo[{i}])' for i in range(4)) + ', [vstat] "v"(vstat)'
    print("#define KD4_IN_KF " + kf)
    print("#define KD4_IN_ADDR " + ad)
    print("#define KD4_IN_DMA " + dm)
    print('#define KD4_IN_MASK [qlo0] "v"(mqlo[0]), [qlo1] "v"(mqlo[1]), [qhi0] "v"(mqhi[0]), [qhi1] "v"(mqhi[1])')
    print('#define KD4_CLOBBER "memory", "scc", ' + ", ".join(f'"s{i}"' for i in range(84, 96)) + ", UAMD_ACC256_CLOBBER, " +
          ", ".join(f'"v{i}"' for i in range(128, 256)))
    print("// clang-format on")


if __name__ == "__main__":
    main()
