#!/usr/bin/env python
"""Where in the backward does each data-parallel collective become runnable, and how much backward is left to hide it?
From a rocprofv3 trace with --kernel-trace --rccl-trace --hip-runtime-trace (rocpd sqlite). usage: rocpd_overlap.py results.db

The host runs far ahead of the GPU (a step is GPU-bound), so the CPU time of an ncclAllReduce call says nothing by itself.
What fixes a collective's place on the GPU timeline is STREAM ORDER: torch records an event on the compute stream at the call
(= after the last kernel launched before it) and RCCL's stream waits for that event. So for every collective call of the last
step:  last kernel-launch API call before it (CPU order) -> that kernel's dispatch (correlation id) -> its END on the GPU
= the moment the bucket is complete and the collective may start;  the compute kernels from there to the optimizer kernel are
what it overlaps with. If RCCL kernels are in the trace (N > 1 ranks) their own intervals are listed too; in a 1-rank group
RCCL elides the in-place collective (nothing to move), and only the ready points can be shown. Markdown on stdout."""
import bisect
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_stats import short  # noqa: E402

COLL = ("ncclAllReduce", "ncclReduceScatter", "ncclAllGather")


def main():
    c = sqlite3.connect(sys.argv[1])
    kern = c.execute(
        "select K.start, K.end, S.display_name, K.queue_id, E.stack_id from rocpd_kernel_dispatch K "
        "join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid "
        "join rocpd_event E on E.id = K.event_id and E.guid = K.guid order by K.start").fetchall()
    names = {}
    recs = []
    for st, en, n, q, corr in kern:
        if n not in names:
            names[n] = short(n)
        recs.append((st, en, names[n], q, corr))
    api = c.execute(
        "select R.start, R.end, S.string, R.id, R.tid from rocpd_region R join rocpd_string S on S.id = R.name_id and S.guid = R.guid "
        "order by R.start").fetchall()
    census = {}
    for r in api:
        if r[2].startswith("nccl"):
            census[r[2]] = census.get(r[2], 0) + 1
    print(f"# Data-parallel collectives on the GPU timeline ({sys.argv[1].rsplit('/', 1)[-1]})\n")
    print("RCCL API calls in the whole trace: " + (", ".join(f"{k} x{v}" for k, v in sorted(census.items())) or "none") + "\n")
    is_coll_k = lambda n: "nccl" in n.lower() or "rccl" in n.lower()          # noqa: E731
    coll_k = [r for r in recs if is_coll_k(r[2])]
    print(f"RCCL kernels in the trace: {len(coll_k)}" + ("" if coll_k else " (1-rank group: the in-place collective moves nothing; RCCL launches no kernel)") + "\n")
    by_corr = {r[4]: r for r in recs}
    # a dispatch's event carries the region id of the API call that launched it (rocpd_event.stack_id; correlation ids are all 0)
    launches = [(a[0], a[3], a[4]) for a in api if a[3] in by_corr]                       # (CPU time, region id, thread)
    if not launches:
        print("(no kernel-launch API records matched to dispatches: run rocprofv3 with --hip-runtime-trace)")
        print("API names seen: " + ", ".join(sorted({a[2] for a in api})[:40]))
        return
    lnames = {}
    for a in api:
        if a[3] in by_corr:
            lnames[a[2]] = lnames.get(a[2], 0) + 1
    print(f"API calls that launched a kernel: {lnames} -> {len(launches)} of {len(recs)} dispatches linked\n")
    opt = [r for r in recs if "adamw" in r[2].lower() or "Adam" in r[2]]
    # the last step on the GPU starts at its embedding lookup (the forward's first kernel) and ends with its last optimizer kernel
    emb = [r for r in recs if "gather_kernel" in r[2] or "mbedding" in r[2] or "index_select" in r[2].lower()]
    if not opt or not emb:
        print("(cannot delimit a step: no optimizer / embedding kernel found)")
        return
    last_opt = opt[-1]
    starts = [r for r in emb if r[0] < last_opt[0]]
    # the embedding kernel that opens the last step: the latest one that is followed by that step's backward (>= 10 ms before the optimizer)
    starts = [r for r in starts if last_opt[0] - r[0] > 10_000_000] or starts
    t_first = starts[-1][0]
    step = [r for r in recs if t_first <= r[0] <= last_opt[1] and not is_coll_k(r[2])]
    first_opt_of_step = min((r for r in opt if r[0] >= t_first), key=lambda r: r[0])
    t0 = step[0][0]
    bwd0 = next((r[0] for r in step if "bwd" in r[2] or "backward" in r[2]), t0)
    per_tid = {}
    for x in launches:
        per_tid.setdefault(x[2], []).append(x)
    rows = []
    for st, en, name, rid, tid in api:
        if name not in COLL:
            continue
        mine = per_tid.get(tid, [])                       # the hook runs on the thread that launches the backward kernels
        i = bisect.bisect_left([x[0] for x in mine], st) - 1
        if i < 0:
            continue
        k = by_corr[mine[i][1]]
        if not (t_first <= k[0] <= last_opt[1]):
            continue
        rows.append((name, k))
    print(f"## Last step: {(last_opt[1] - t0) / 1e6:.1f} ms on the GPU, backward from {(bwd0 - t0) / 1e6:.1f} ms, first optimizer kernel at "
          f"{(first_opt_of_step[0] - t0) / 1e6:.1f} ms; {len(rows)} collectives\n")
    print("| # | call | runnable at (ms into the step) = end of | compute kernels queued behind that point until the optimizer | of which ms |")
    print("|---|---|---|---|---|")
    for i, (name, k) in enumerate(rows):
        behind = [r for r in step if r[0] >= k[1] and r[0] < first_opt_of_step[0]]
        ms = sum(r[1] - r[0] for r in behind) / 1e6
        print(f"| {i} | {name} | {(k[1] - t0) / 1e6:.2f} = `{k[2][:48]}` | {len(behind)} | {ms:.1f} |")
    if rows:
        last_ready = max(k[1] for _, k in rows)
        tail = [(n, k) for n, k in rows if first_opt_of_step[0] - k[1] < 2_000_000]
        print(f"\n{len(rows) - len(tail)} of {len(rows)} collectives become runnable with more than 2 ms of compute kernels still queued behind them; "
              f"the last one is runnable {(first_opt_of_step[0] - last_ready) / 1e6:.2f} ms before the first optimizer kernel starts.")
    if coll_k:
        print("\n## RCCL kernels of the last step\n")
        print("| # | kernel | queue | start (ms into the step) | duration us | compute kernels running concurrently (overlap us) |")
        print("|---|---|---|---|---|---|")
        for i, (st, en, n, q, _) in enumerate([r for r in coll_k if t_first <= r[0] <= last_opt[1]]):
            over = sorted(((min(en, r[1]) - max(st, r[0])) / 1e3, r[2]) for r in step if r[1] > st and r[0] < en)[::-1][:4]
            print(f"| {i} | {n[:50]} | {q} | {(st - t0) / 1e6:.2f} | {(en - st) / 1e3:.0f} | " + ("; ".join(f"{n2[:30]} ({o:.0f})" for o, n2 in over) or "none") + " |")


if __name__ == "__main__":
    main()
