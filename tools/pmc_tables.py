#!/usr/bin/env python
"""Markdown tables from the tools/pmc_summary.py outputs of one bench.py PMC campaign.

  pmc_tables.py mfma  <mfma.txt>                      -> MFMA-busy / effective clock / fraction of the dense bf16 peak
  pmc_tables.py hbm   <fetch.txt> <write.txt> [T]     -> HBM-side bytes per launch / duration = GB/s, fraction of 8 TB/s,
                                                         beside the ALGORITHMIC bytes of SURVEY 8(d) at T tokens (8192)

Counters and corrections are the ones MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KiB from separate
passes, traffic = 2 * FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE under-reports wide streaming reads by 2x);
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024
SIMDs); fraction of the 2.5 PFLOP/s dense peak = busy * clock / 2.4 GHz. Durations are the PMC pass's own (counter
collection serialises dispatches; the kernel-trace run of the same command is the timing reference)."""
import re
import sys

HBM_PEAK = 8.0e12
N_XCD, N_SIMD, F_PEAK_GHZ = 8, 1024, 2.4


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?) avg_dur_us=([\d.]+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {"avg_dur_us": float(m.group(2))}
            continue
        m = re.match(r"\s+(\S+)\s+mean/dispatch = ([\d.e+-]+)\s+\(n=(\d+)\)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
            out[cur]["n"] = int(m.group(3))
    return out


def algorithmic_bytes(name, T, H=4096, I=14336, V=128256, Hq=32, Hk=8, D=128):
    """SURVEY 8(d) per-call figures for the kernels of the Llama-3-8B step (bf16); None where the table has no row."""
    if name.startswith("rms_fwd_rb"):
        # ",201" = the add-fused instance (reads x and the residual, writes h and y): 4 row passes instead of 2
        return (4 if name.rstrip(">").endswith("1") else 2) * T * H * 2 + H * 2 + T * 4
    if name.startswith("rms_bwd_rb"):
        return (4 if name.rstrip(">").endswith("1") else 3) * T * H * 2 + H * 2 + T * 4
    if name.startswith("rope_vec"):
        return 2 * T * (Hq + Hk) * D * 2 + 2 * T * (D // 2) * 2
    if name.startswith("glu_fwd"):
        return 3 * T * I * 2
    if name.startswith("glu_xa"):
        # glu_xa_kernel<T, ACT, NS, NT, KS, PD> printed as "<bf16,0NS...>": NS = 1 is the forward twin (3 tensor passes), 2 the backward
        m = re.search(r"<\w+,\d(\d)", name)
        return (3 if m and m.group(1) == "1" else 6) * T * I * 2
    if name.startswith("glu_bwd"):
        return 6 * T * I * 2
    if name.startswith("ce_fwd"):
        return None          # row chunks: the chunk height is not in the name
    return None


def main():
    mode = sys.argv[1]
    if mode == "mfma":
        d = parse(sys.argv[2])
        rows = []
        for k, v in d.items():
            busy, act = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
            if busy <= 0 or act <= 0:
                continue
            clk = act / N_XCD / (v["avg_dur_us"] * 1e3)          # cycles per ns = GHz
            frac_busy = busy / (act / N_XCD * N_SIMD)
            rows.append((v["avg_dur_us"], k, clk, frac_busy, frac_busy * clk / F_PEAK_GHZ, v.get("n", 0)))
        print("| kernel | launches | avg launch (us) | effective clock (GHz) | MFMA-busy | x clock/2.4 = fraction of 2.5 PFLOP/s |")
        print("|---|---|---|---|---|---|")
        for dur, k, clk, fb, fp, n in sorted(rows, reverse=True):
            print(f"| `{k}` | {n} | {dur:.1f} | {clk:.2f} | {100 * fb:.1f} % | {100 * fp:.1f} % |")
    elif mode == "hbm":
        f, w = parse(sys.argv[2]), parse(sys.argv[3])
        T = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
        print("| kernel | launches | avg launch (us) | HBM-side bytes / launch (2*FETCH+WRITE) | GB/s | of 8 TB/s | algorithmic bytes | traffic / algorithmic |")
        print("|---|---|---|---|---|---|---|---|")
        rows = []
        for k in set(f) & set(w):
            if "FETCH_SIZE" not in f[k] or "WRITE_SIZE" not in w[k]:
                continue
            by = 2 * f[k]["FETCH_SIZE"] * 1024 + w[k]["WRITE_SIZE"] * 1024
            dur = 0.5 * (f[k]["avg_dur_us"] + w[k]["avg_dur_us"])
            if dur < 8 or by < 1e6:
                continue
            rows.append((dur * f[k].get("n", 1), k, f[k].get("n", 0), dur, by))
        for _, k, n, dur, by in sorted(rows, reverse=True):
            alg = algorithmic_bytes(k, T)
            gbs = by / (dur * 1e-6) / 1e9
            print(f"| `{k}` | {n} | {dur:.1f} | {by / 1e6:.1f} MB | {gbs:.0f} | {100 * gbs * 1e9 / HBM_PEAK:.0f} % | "
                  + (f"{alg / 1e6:.1f} MB | {by / alg:.2f} |" if alg else "- | - |"))
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
