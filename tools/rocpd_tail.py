#!/usr/bin/env python
"""The last N kernels of a rocprofv3 kernel trace (rocpd sqlite) in start order: start offset, duration, END offset, queue, short
name -- for traces where launches of two streams overlap (the decode engine's early-started gate|up), which a
start/duration/gap list cannot show. usage: rocpd_tail.py results.db [N=60] [skip_last=0]"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocpd_stats import short  # noqa: E402


def main():
    c = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    q = "K.queue_id" if "queue_id" in cols else "0"
    rows = c.execute(f"select K.start, K.end, {q}, S.display_name from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S "
                     "on S.id=K.kernel_id and S.guid=K.guid order by K.start").fetchall()
    rows = rows[len(rows) - n - skip:len(rows) - skip]
    t0 = rows[0][0]
    print("start_us,dur_us,end_us,queue,name")
    for st, en, qu, name in rows:
        print(f"{(st - t0) / 1e3:.1f},{(en - st) / 1e3:.1f},{(en - t0) / 1e3:.1f},{qu},{short(name)}")


if __name__ == "__main__":
    main()
