#!/usr/bin/env python3
"""Per-kernel (name + grid) averages of rocprofv3 --pmc counters over several PMC passes of the
same command, merged into one JSON: what the kernels of the dependent chain behind the top-level
pass wait for. Dev tool.

usage: tools/chain_counters.py out.json pass1.db pass2.db ...

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SE-instance. Derived per
kernel: wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked at s_waitcnt / barrier),
wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stall: dependency / pipe), active =
SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)."""
import json
import sqlite3
import sys


def per_kernel(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks, pe, pi = (T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"),
                      T("rocpd_info_pmc"))
    q = f"""select s.kernel_name, d.grid_size_x, d.grid_size_y, p.name, d.dispatch_id, sum(e.value),
                   d.end - d.start
            from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id
            join {pi} p on e.pmc_id = p.id group by 1, 2, 3, 4, 5"""
    out = {}
    for name, gx, gy, ctr, _disp, val, dur in c.execute(q):
        key = "%s grid=%dx%d" % (name.split("(")[0].replace(".kd", ""), gx, gy)
        out.setdefault(key, {}).setdefault(ctr, []).append((val, dur))
    res = {}
    for key, ctrs in out.items():
        res[key] = {}
        for ctr, v in ctrs.items():
            v = v[1:] if len(v) > 1 else v  # first dispatch: warm-up
            res[key][ctr] = sum(x[0] for x in v) / len(v)
            res[key]["_dur_us_under_pmc"] = sum(x[1] for x in v) / len(v) / 1e3
            res[key]["_launches"] = len(v)
    return res


def main():
    merged = {}
    for db in sys.argv[2:]:
        for key, ctrs in per_kernel(db).items():
            merged.setdefault(key, {}).update(ctrs)
    for key, c in merged.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n, src in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"),
                           ("active", "SQ_ACTIVE_INST_ANY"), ("active_valu", "SQ_ACTIVE_INST_VALU"),
                           ("active_lds", "SQ_ACTIVE_INST_LDS"), ("active_vmem", "SQ_ACTIVE_INST_VMEM")):
                if src in c:
                    c["frac_" + n] = round(c[src] / wc, 4)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            c["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "SQ_WAVES" in c and c["SQ_WAVES"]:
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                      "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
                if n in c:
                    c[n + "_per_wave"] = round(c[n] / c["SQ_WAVES"], 1)
    keep = {k: v for k, v in merged.items() if "mgh" in k}
    json.dump({"_about": __doc__.strip(), "kernels": keep}, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    for k in sorted(keep):
        c = keep[k]
        print("%-78s dur %7.1f us  wait_any %.2f wait_inst %.2f active %.2f (valu %.2f lds %.2f vmem %.2f)  l2hit %s" % (
            k[-78:], c.get("_dur_us_under_pmc", 0), c.get("frac_wait_any", -1), c.get("frac_wait_inst", -1),
            c.get("frac_active", -1), c.get("frac_active_valu", -1), c.get("frac_active_lds", -1),
            c.get("frac_active_vmem", -1), c.get("l2_hit", "-")))


if __name__ == "__main__":
    main()
