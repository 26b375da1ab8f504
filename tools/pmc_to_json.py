#!/usr/bin/env python
"""profiles/pmc_traffic.json from the two tools/pmc_summary.py outputs of tools/gpu_pmc_bench.sh (FETCH_SIZE pass,
WRITE_SIZE pass). traffic_bytes = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 per dispatch (the gfx950 correction
of MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide streaming read).
usage: pmc_to_json.py fetch.txt write.txt > profiles/pmc_traffic.json"""
import json
import re
import sys


def parse(path, counter):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?) avg_dur_us=([\d.]+)", line)
        if m:
            cur = m.group(1)
            out.setdefault(cur, {})["avg_dur_us"] = float(m.group(2))
            continue
        m = re.match(rf"\s+{counter}\s+mean/dispatch = ([\d.e+]+)\s+\(n=(\d+)\)", line)
        if m and cur:
            out[cur][counter] = float(m.group(1))
            out[cur]["n"] = int(m.group(2))
    return out


f, w = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB per dispatch, mean over the dispatches of "
                "`bench.py --steps 2 --warmup 1`), tools/gpu_pmc_bench.sh + tools/pmc_to_json.py. traffic_bytes = "
                "2*FETCH_SIZE*1024 + WRITE_SIZE*1024: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950 "
                "(MI355X_MICROARCH.md, HBM section). Counts L2<->fabric requests, Infinity-Cache hits included."}
for k in sorted(set(f) | set(w)):
    fe, wr = f.get(k, {}).get("FETCH_SIZE"), w.get(k, {}).get("WRITE_SIZE")
    if fe is None or wr is None:
        continue
    res[k] = {"fetch_kib": fe, "write_kib": wr, "dispatches": f[k].get("n", 0), "avg_dur_us": f[k].get("avg_dur_us"),
              "traffic_bytes": int(2 * fe * 1024 + wr * 1024)}
json.dump(res, sys.stdout, indent=1)
