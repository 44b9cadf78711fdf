#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite db. Dev tool."""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
def T(prefix):
    return [t for t in tabs if t.startswith(prefix)][0]
kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
cols = [r[1] for r in c.execute(f"pragma table_info({pe})")]
q = f"""select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, p.name, avg(e.value), count(*)
from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id
join {pi} p on e.pmc_id = p.id group by 1,2,3,4,5 order by 1,2,3,4,5"""
for r in c.execute(q):
    if 'mgh' in r[0]:
        print("%-60s grid=(%d,%d,%d) %-22s avg=%.4g n=%d" % (r[0].split('(')[0][:60], r[1], r[2], r[3], r[4], r[5], r[6]))
