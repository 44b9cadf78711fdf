#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per kernel name and grid, launch count and
average duration, skipping the first `--skip` launches of each (warm-up). Dev tool."""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--skip", type=int, default=3)
ap.add_argument("--csv", default=None)
a = ap.parse_args()
c = sqlite3.connect(a.db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"""select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z,
    d.start, d.end, d.group_segment_size, s.arch_vgpr_count, s.sgpr_count
    from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"""))
agg = {}
for name, gx, gy, gz, st, en, lds, vg, sg in rows:
    key = (name.split("(")[0].replace(".kd", ""), gx, gy, gz, lds, vg, sg)
    agg.setdefault(key, []).append((en - st) / 1000.0)
tot = {}
lines = []
for key, d in agg.items():
    dd = d[a.skip:] if len(d) > a.skip else d
    avg = sum(dd) / len(dd)
    tot[key[0]] = tot.get(key[0], 0) + avg
    lines.append((key, len(d), avg))
lines.sort(key=lambda x: (x[0][0], -x[2]))
out = ["kernel,grid_x,grid_y,grid_z,lds_bytes,vgpr,sgpr,launches,avg_us"]
for key, n, avg in lines:
    out.append("%s,%d,%d,%d,%d,%d,%d,%d,%.2f" % (*key, n, avg))
out.append("")
out.append("kernel,sum_of_avg_us_over_grids (= per step)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    out.append("%s,%.2f" % (k, v))
out.append("TOTAL,%.2f" % sum(tot.values()))
txt = "\n".join(out)
if a.csv:
    open(a.csv, "w").write(txt + "\n")
print(txt)
