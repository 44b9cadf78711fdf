#!/bin/bash
# rocprofv3 kernel trace of mgh_compress / mgh_decompress (device-resident, 512^3 f32): every launch of
# the LAST compress and the LAST decompress call with start, duration and the gap in front of it.
# Usage (through gpurun): tools/exp_hl_trace.sh TAG
TAG=${1:-hl}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -- python3 $GRAFT_REPO_ROOT/tools/exp_e2e_out.py > /tmp/prof_$TAG.log 2>&1
F=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
def show(first_pat, title):
    last = max(i for i, n in enumerate(names) if first_pat in n)
    t0 = int(rows[last]["Start_Timestamp"]); prev_end = t0
    print("==", title)
    for r in rows[last:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (s - t0) > 4e6: break
        print("%9.1f  dur %7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:80]))
        prev_end = max(prev_end, e)
# the last compress call starts with the zeroing launch in front of absmax; the last decompress with the record pieces
show("k_zero_words", "last mgh_compress")
show("k_record_pieces", "last mgh_decompress (from its record-pieces launch)")
PY
tail -3 /tmp/prof_$TAG.log
