#!/usr/bin/env python
"""The LAST training step of a rocprofv3 kernel trace (rocpd sqlite) as an ordered list: start offset, duration, gap to
the previous kernel's end, short kernel name. usage: rocpd_sequence.py results.db [n_steps] > step_sequence.csv
The step boundary is found from the periodicity of attn_fwd_kernel launches (32 per step for Llama-3-8B): the last
step = everything after the end of the backward that precedes the last block of forward launches. Summary lines at
the end: kernel time, idle gaps (total and the 10 largest), per-name totals of everything shorter than 20 us."""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_stats import short  # noqa: E402


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute(
        "select K.start, K.end, S.display_name from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S "
        "on S.id=K.kernel_id and S.guid=K.guid order by K.start").fetchall()
    names = {}
    recs = []
    for st, en, n in rows:
        if n not in names:
            names[n] = short(n)
        recs.append((st, en, names[n]))
    # last optimizer launch before the final step's first forward kernel: steps end with the (multi-tensor or flat) AdamW
    opt = [i for i, r in enumerate(recs) if "Adam" in r[2] or "adamw" in r[2]]
    if len(opt) < 2:
        first = max(0, len(recs) - 4000)
    else:
        # walk back from the last optimizer launch to the previous block of optimizer launches
        last = opt[-1]
        j = len(opt) - 1
        while j > 0 and opt[j] - opt[j - 1] < 50:
            j -= 1
        first = opt[j - 1] + 1 if j > 0 else 0
        recs = recs[:last + 1]
    step = recs[first:]
    t0 = step[0][0]
    print("start_us,dur_us,gap_us,name")
    prev_end = step[0][0]
    ksum = gsum = 0.0
    gaps, small = [], {}
    for st, en, n in step:
        gap = max(0, st - prev_end) / 1e3
        dur = (en - st) / 1e3
        print(f"{(st - t0) / 1e3:.1f},{dur:.1f},{gap:.1f},{n}")
        ksum += dur
        gsum += gap
        gaps.append((gap, n))
        if dur < 20:
            a = small.setdefault(n, [0, 0.0])
            a[0] += 1
            a[1] += dur
        prev_end = max(prev_end, en)
    span = (prev_end - t0) / 1e3
    print(f"# kernels {len(step)}, span {span / 1e3:.2f} ms, kernel time {ksum / 1e3:.2f} ms, idle gaps {gsum / 1e3:.2f} ms")
    print("# largest gaps (us, before kernel): " + "; ".join(f"{g:.0f} {n}" for g, n in sorted(gaps, reverse=True)[:10]))
    for n, (cnt, tot) in sorted(small.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"# short kernel {n}: {cnt} launches, {tot / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
