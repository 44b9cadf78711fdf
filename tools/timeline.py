#!/usr/bin/env python3
"""Timeline of the LAST step in a rocprofv3 (rocpd sqlite) kernel trace: start / end / duration of
every launch relative to the step's first kernel (k_absmax or k_make_qparams), so that overlap of
launches on different streams and the gaps between dependent launches can be read off. Dev tool.
usage: tools/timeline.py trace.db [--first k_absmax] [--steps-back 1]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--first", default="k_absmax")
ap.add_argument("--steps-back", type=int, default=1)
a = ap.parse_args()
c = sqlite3.connect(a.db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"""select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.queue_id
    from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"""))
starts = [i for i, r in enumerate(rows) if a.first in r[0]]
if len(starts) <= a.steps_back:
    # (an L2 norm starts the step with k_sqsum instead of k_absmax; a given norm with k_make_qparams)
    for alt in ("k_sqsum", "k_make_qparams"):
        starts = [i for i, r in enumerate(rows) if alt in r[0]]
        if len(starts) > a.steps_back:
            break
i0 = starts[-1 - a.steps_back]
i1 = starts[-a.steps_back] if a.steps_back > 0 else len(rows)
t0 = rows[i0][1]
prev_end = t0
for name, st, en, gx, gy, q in rows[i0:i1]:
    short = name.split("(")[0].replace("_ZN3mgh", "").replace(".kd", "")[:60]
    print("%8.1f %8.1f  dur %7.1f  gap %6.1f  q%-3s grid %7d x %-4d %s" % (
        (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, q, gx, gy, short))
    prev_end = max(prev_end, en)
print("step span %.1f us" % ((max(r[2] for r in rows[i0:i1]) - t0) / 1e3))
