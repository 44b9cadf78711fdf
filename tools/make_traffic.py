#!/usr/bin/env python3
"""profiles/traffic_512cube_f32.json from two rocprofv3 PMC passes of the bench workload
(FETCH_SIZE and WRITE_SIZE must be collected in separate runs, kernel-trace only):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/fetch -- python3 bench.py --steps 3 --warmup 1 --only-step
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/write -- python3 bench.py --steps 3 --warmup 1 --only-step
  python tools/make_traffic.py out/fetch/*/*.db out/write/*/*.db [raw_out.json [traffic_file_name]]

(traffic_file_name: default traffic_512cube_f32.json; e.g. traffic_1024cube_f32.json for
`bench.py --config 1024f32`.)

The file records the hash of the kernel sources it was taken on; bench.py refuses it for any
other sources. Units: KiB per launch (averages over the dispatches of the biggest grid of each
kernel). read_correction: /opt/skills/guides/MI355X_MICROARCH.md, HBM section -- on gfx950
FETCH_SIZE reports half of the bytes of coalesced streaming loads. Calibrated here on known byte
counts (tools/micro/pmc_calib.hip, tools/pmc_calib_summary.py): 512 MiB read with 4-byte-per-lane
AND with 16-byte-per-lane contiguous loads both count 262 154 KiB (x 2), 1024 MiB written with
8-byte stores, plain or nontemporal, count 1 048 576 ... 1 048 850 KiB (x 1).
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    q = f"""select s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z, d.dispatch_id, sum(e.value)
            from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id
            join {pi} p on e.pmc_id = p.id where p.name = ? group by 1, 2, 3"""
    out = {}
    for name, grid, _disp, val in c.execute(q, (counter,)):
        out.setdefault((name.split("(")[0], grid), []).append(val)
    return out


def tmode_is(symbol, tmode):
    """k_level_fused2<T, OUT, TC, TF, RCH, FACES, TMODE, AGG>: is TMODE (the last integer template
    argument of the mangled symbol, in front of the bool AGG since round 5) = tmode?"""
    return any(("ELi%d%sEEEv" % (tmode, tail)) in symbol for tail in ("", "ELb0", "ELb1"))


def biggest(stats, needle):
    keys = [k for k in stats if needle in k[0]]
    if not keys:
        return None, 0
    k = max(keys, key=lambda kk: kk[1])
    v = stats[k]
    v = v[1:] if len(v) > 1 else v  # first dispatch: warm-up
    return sum(v) / len(v), len(v)


def main():
    from bench import source_hash
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_about": __doc__.strip(), "source_hash": source_hash()}
    for label, needle, corr in (("level_fused_q", "k_level_fused2", 2.0), ("absmax", "k_absmax", 2.0),
                                ("sqsum", "k_sqsum", 2.0)):
        pick = (lambda st: {k: v for k, v in st.items() if tmode_is(k[0], 0)}) if needle == "k_level_fused2" else (lambda st: st)
        f, nf = biggest(pick(fetch), needle)
        w, nw = biggest(pick(write), needle)
        if f is None or w is None:
            continue
        out[label] = {"fetch_kib": round(f, 1), "write_kib": round(w, 1), "read_correction": corr,
                      "dispatches": [nf, nw]}
    # D = 4: the level kernel under the names the library profiles it with, per t-slice parity
    # (template argument TMODE = 1 / 2 at the end of the symbol; absent from a 3-D run)
    for label, tmode in (("level4_even", 1), ("level4_odd", 2)):
        sub_f = {k: v for k, v in fetch.items() if "k_level_fused2" in k[0] and tmode_is(k[0], tmode)}
        sub_w = {k: v for k, v in write.items() if "k_level_fused2" in k[0] and tmode_is(k[0], tmode)}
        f, nf = biggest(sub_f, "k_level_fused2")
        w, nw = biggest(sub_w, "k_level_fused2")
        if f is not None and w is not None:
            out[label] = {"fetch_kib": round(f, 1), "write_kib": round(w, 1), "read_correction": 2.0,
                          "dispatches": [nf, nw]}
    # the Thomas solves: ALL k_ipk* launches of one step together (the kernels and their launch
    # counts differ between configurations: LDS-staged, streaming, LDS-DMA, plane-fused, ranges of
    # r-planes at 1024^3), against ipk_f + ipk_c + ipk_r of bench.py:algorithmic_bytes_per_step
    steps_f = max(len(v) for k, v in fetch.items() if "k_make_qparams" in k[0]) if any("k_make_qparams" in k[0] for k in fetch) else 1
    steps_w = max(len(v) for k, v in write.items() if "k_make_qparams" in k[0]) if any("k_make_qparams" in k[0] for k in write) else 1
    tf = sum(sum(v) for k, v in fetch.items() if "k_ipk" in k[0] or "k_tsolve" in k[0]) / steps_f
    tw = sum(sum(v) for k, v in write.items() if "k_ipk" in k[0] or "k_tsolve" in k[0]) / steps_w
    out["ipk_all_per_step"] = {"fetch_kib": round(tf, 1), "write_kib": round(tw, 1), "read_correction": 2.0,
                               "steps": [steps_f, steps_w],
                               "what": "every Thomas-solve launch of one step, summed"}
    name = sys.argv[4] if len(sys.argv) > 4 else "traffic_512cube_f32.json"
    json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    if len(sys.argv) > 3:
        raw = {"fetch": [{"kernel": k[0][:90], "grid_threads": k[1], "avg_kib": sum(v) / len(v), "n": len(v)}
                         for k, v in sorted(fetch.items()) if "mgh" in k[0]],
               "write": [{"kernel": k[0][:90], "grid_threads": k[1], "avg_kib": sum(v) / len(v), "n": len(v)}
                         for k, v in sorted(write.items()) if "mgh" in k[0]],
               "source_hash": out["source_hash"]}
        json.dump(raw, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "_about"}, indent=1))


if __name__ == "__main__":
    main()
