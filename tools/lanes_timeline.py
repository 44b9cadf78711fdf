#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 (rocpd sqlite) kernel + memory-copy trace: for the last `--window-ms`
of the trace, every launch with its queue, plus per-queue busy time and the time two queues were
busy at once. Dev tool (the two-lane subdomain pipeline of mgh_compress / mgh_decompress).
usage: tools/lanes_timeline.py trace.db [--window-ms 250] [--min-us 20]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--window-ms", type=float, default=250.0)
ap.add_argument("--min-us", type=float, default=20.0)
a = ap.parse_args()
c = sqlite3.connect(a.db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"""select s.kernel_name, d.start, d.end, d.queue_id
    from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"""))
end = max(r[2] for r in rows)
t0 = end - a.window_ms * 1e6
rows = [r for r in rows if r[1] >= t0]
t0 = rows[0][1]
queues = sorted({r[3] for r in rows})
print("queues:", queues)
for name, st, en, q in rows:
    if (en - st) / 1e3 < a.min_us:
        continue
    short = name.split("(")[0].replace("_ZN3mgh", "").replace(".kd", "")[:70]
    print("%9.3f %9.3f ms  dur %8.1f us  %s q%-3s %s" % ((st - t0) / 1e6, (en - t0) / 1e6, (en - st) / 1e3,
                                                     " " * (12 * queues.index(q)), q, short))
# busy time per queue and pairwise overlap
ev = []
for _, st, en, q in rows:
    ev.append((st, 1, q))
    ev.append((en, -1, q))
ev.sort()
active = {q: 0 for q in queues}
last = ev[0][0]
busy = {q: 0 for q in queues}
both = 0
anyb = 0
for t, d, q in ev:
    n = sum(1 for v in active.values() if v > 0)
    for qq, v in active.items():
        if v > 0:
            busy[qq] += t - last
    if n >= 2:
        both += t - last
    if n >= 1:
        anyb += t - last
    last = t
    active[q] += d
span = (rows[-1][2] - t0) / 1e6
print("span %.2f ms; busy per queue (ms):" % span, {q: round(v / 1e6, 2) for q, v in busy.items()},
      "; >= 2 queues busy %.2f ms; any busy %.2f ms" % (both / 1e6, anyb / 1e6))
