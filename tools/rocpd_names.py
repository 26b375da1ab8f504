#!/usr/bin/env python
"""Full kernel names + grid/workgroup/LDS/VGPR of every distinct kernel in a rocprofv3 rocpd database."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute(
    "select S.display_name, count(*), avg(K.end-K.start), max(K.workgroup_size_x), max(K.grid_size_x), "
    "max(S.arch_vgpr_count), max(S.accum_vgpr_count), max(K.group_segment_size) "
    "from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id=K.kernel_id and S.guid=K.guid "
    "group by S.display_name order by 3 desc").fetchall()
for n, calls, avg, wg, grid, vg, ag, lds in rows:
    print(f"calls={calls} avg_us={avg / 1e3:.1f} wg={wg} grid={grid} vgpr={vg} agpr={ag} lds={lds}\n    {n}")
