#!/usr/bin/env python
"""Do the data-parallel collectives run CONCURRENTLY with backward kernels? From a rocprofv3 kernel trace (rocpd sqlite):
every RCCL kernel (name contains nccl / rccl) of the last training step with its queue, its start / duration, and the
compute kernels of OTHER queues whose execution interval intersects it (name, overlap in us). Markdown on stdout.
usage: rocpd_overlap.py results.db"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_stats import short  # noqa: E402


def api_issue_points(c, recs):
    """RCCL API calls (rocprofv3 --rccl-trace -> rocpd_region) of the last step and what the GPU was executing at the
    moment each one was issued: a collective issued while backward kernels of later buckets are still running (and long
    before the optimizer kernel) is what "overlapped with backward" means on the host side."""
    try:
        rows = c.execute("select R.start, R.end, S.string from rocpd_region R join rocpd_string S on S.id = R.name_id "
                         "where S.string like 'nccl%' order by R.start").fetchall()
    except sqlite3.Error as ex:
        print(f"(no RCCL API regions: {ex})")
        return
    census = {}
    for r in rows:
        census[r[2]] = census.get(r[2], 0) + 1
    print("RCCL API calls in the whole trace: " + ", ".join(f"{k} x{v}" for k, v in sorted(census.items())) + "\n")
    rows = [r for r in rows if r[2] in ("ncclAllReduce", "ncclReduceScatter", "ncclAllGather")]
    if not rows:
        print("(no ncclAllReduce / ncclReduceScatter / ncclAllGather API calls in the trace)")
        return
    opt = [r[0] for r in recs if "adamw" in r[2].lower() or "Adam" in r[2]]
    t_first = opt[-2] if len(opt) >= 2 else recs[0][0]
    t_last = opt[-1] if opt else recs[-1][1]
    step = [r for r in recs if t_first < r[0] <= t_last]
    calls = [r for r in rows if t_first < r[0] <= t_last + 5_000_000]
    if not step or not calls:
        return
    bwd0 = next((r[0] for r in step if "bwd" in r[2] or "backward" in r[2]), step[0][0])
    t0, t1 = step[0][0], step[-1][1]
    print(f"## RCCL API calls of the last step ({len(calls)}), against the GPU timeline of that step "
          f"(step = {(t1 - t0) / 1e6:.1f} ms, backward starts at {(bwd0 - t0) / 1e6:.1f} ms, optimizer kernel at {(t_last - t0) / 1e6:.1f} ms)\n")
    print("| # | call | issued at (ms into the step) | kernel executing on the GPU at that moment | GPU work still queued behind it (ms until the optimizer kernel) |")
    print("|---|---|---|---|---|")
    for i, (st, en, name) in enumerate(calls):
        cur = next((r for r in step if r[0] <= st < r[1]), None)
        if cur is None:
            nxt = next((r for r in step if r[0] >= st), None)
            what = f"(idle; next: {nxt[2]})" if nxt else "(after the step's last kernel)"
        else:
            what = cur[2]
        print(f"| {i} | {name} | {(st - t0) / 1e6:.2f} | {what[:70]} | {max(0.0, (t_last - st) / 1e6):.2f} |")
    print()


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    scol = "stream_id" if "stream_id" in cols else "0"
    rows = c.execute(
        f"select K.start, K.end, S.display_name, {('K.' + qcol) if qcol else '0'}, {('K.' + scol) if scol != '0' else '0'} "
        "from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id=K.kernel_id and S.guid=K.guid order by K.start").fetchall()
    names = {}
    recs = []
    for st, en, n, q, s in rows:
        if n not in names:
            names[n] = short(n)
        recs.append((st, en, names[n], q, s))
    is_coll = lambda n: "nccl" in n.lower() or "rccl" in n.lower()          # noqa: E731
    coll = [r for r in recs if is_coll(r[2])]
    print(f"# RCCL kernels vs compute kernels ({sys.argv[1].rsplit('/', 1)[-1]}; dispatch columns: {', '.join(cols)})")
    print(f"\n{len(coll)} collective kernels in the trace, queues {sorted({r[3] for r in coll})}; compute queues "
          f"{sorted({r[3] for r in recs if not is_coll(r[2])})}\n")
    api_issue_points(c, recs)
    if not coll:
        print("\nNo RCCL kernel in the trace: in a 1-rank group RCCL elides the in-place all-reduce / reduce-scatter / all-gather "
              "(nothing to move), so on ONE GPU only the ISSUE points above can be observed.")
        return
    # the last step: collectives after the second-to-last optimizer launch
    opt = [r[0] for r in recs if "adamw" in r[2].lower() or "Adam" in r[2]]
    t_first = opt[-2] if len(opt) >= 2 else recs[0][0]
    last = [r for r in coll if r[0] > t_first]
    t0 = last[0][0] if last else 0
    print("| # | collective kernel | queue | start (us, from the step's first collective) | duration us | concurrent compute kernels on "
          "other queues (overlap us) | overlapped share |")
    print("|---|---|---|---|---|---|---|")
    tot_d = tot_o = 0.0
    for i, (st, en, n, q, s) in enumerate(last):
        over = []
        covered = 0.0
        for st2, en2, n2, q2, s2 in recs:
            if en2 <= st or st2 >= en or is_coll(n2) or (q2 == q and qcol):
                continue
            o = (min(en, en2) - max(st, st2)) / 1e3
            over.append((o, n2))
            covered += o
        d = (en - st) / 1e3
        covered = min(covered, d)
        tot_d += d
        tot_o += covered
        txt = "; ".join(f"{n2} ({o:.0f})" for o, n2 in sorted(over, reverse=True)[:4]) or "-- none --"
        print(f"| {i} | {n[:60]} | {q} | {(st - t0) / 1e3:.0f} | {d:.1f} | {txt} | {100 * covered / max(d, 1e-9):.0f} % |")
    print(f"\nlast step: {len(last)} collective kernels, {tot_d:.0f} us in total, {tot_o:.0f} us of that while a compute kernel of "
          f"another queue was running ({100 * tot_o / max(tot_d, 1e-9):.0f} %).")


if __name__ == "__main__":
    main()
