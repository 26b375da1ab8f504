#!/usr/bin/env python
"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd database. usage: pmc_summary.py results.db [substr]"""
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__file__))
from rocpd_stats import short  # noqa: E402

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute(
    "select S.display_name, K.dispatch_id, P.name, E.value, (K.end-K.start) "
    "from rocpd_pmc_event E join rocpd_info_pmc P on P.id=E.pmc_id and P.guid=E.guid "
    "join rocpd_kernel_dispatch K on K.event_id=E.event_id and K.guid=E.guid "
    "join rocpd_info_kernel_symbol S on S.id=K.kernel_id and S.guid=K.guid").fetchall()
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for name, did, pmc, val, d in rows:
    k = short(name)
    if flt and flt not in k:
        continue
    acc[k][pmc].append((did, val))
    dur[k].append(d)
for k, pm in acc.items():
    print(k, f"avg_dur_us={sum(dur[k]) / len(dur[k]) / 1e3:.1f}")
    for pmc, vals in sorted(pm.items()):
        per = defaultdict(float)
        for did, v in vals:
            per[did] += v
        vs = list(per.values())
        print(f"    {pmc:32s} mean/dispatch = {sum(vs) / len(vs):.4g}  (n={len(vs)})")
