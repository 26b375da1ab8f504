#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / average /
share, the same table `--stats` prints as kernel_stats.csv. usage: rocpd_stats.py results.db [--pmc]"""
import re
import sqlite3
import subprocess
import sys
import os

CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else "c++filt"


def short(name):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:       # our kernels (anonymous namespace); llvm-cxxfilt does not know the DF16b (__bf16) mangling
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        dt = "bf16" if "DF16b" in rest else ("f16" if "DF16_" in rest else ("f32" if rest.startswith("If") else ""))
        flags = "".join(re.findall(r"L[bi](\d+)E", rest))
        return f"{base}<{dt}{',' + flags if flags else ''}>"
    if name.startswith("_Z"):
        try:
            name = subprocess.run([CXXFILT, name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "hipBLASLt " + name[:19] + (" MT" + m.group(1) if m else "")
    m = re.match(r"at::native::(\w+)<", name)
    if m:
        inner = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+(?:Functor|_kernel_cuda|Ops)\w*)", name[len(m.group(0)):])
        return "torch " + m.group(1) + (":" + inner.group(1) if inner else "")
    # keep template args of our own kernels, drop the argument list
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out)[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute(
        "select S.display_name, count(*), sum(K.end-K.start), min(K.end-K.start), max(K.end-K.start), "
        "max(S.arch_vgpr_count), max(S.accum_vgpr_count), max(S.sgpr_count), max(K.group_segment_size) "
        "from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id=K.kernel_id and S.guid=K.guid "
        "group by S.display_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    agg = {}
    for n, calls, tot, mn, mx, vg, ag, sg, lds in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0, 1 << 62, 0, vg, ag, sg, lds])
        a[0] += calls; a[1] += tot; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPR,AGPR,SGPR,LDS")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{k}\",{a[0]},{a[1]},{a[1] / a[0]:.0f},{100.0 * a[1] / total:.3f},{a[2]},{a[3]},{a[4]},{a[5]},{a[6]},{a[7]}")


if __name__ == "__main__":
    main()
