#!/bin/bash
# rocprofv3 kernel trace of the 5-D step: every launch of the generic N-D path with its duration
# (tools/exp_5d_profile.py is the workload). Usage: tools/exp_5d_trace.sh TAG [shape]
TAG=${1:-5d}; SHAPE=${2:-8x8x64x64x64}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -- python3 $GRAFT_REPO_ROOT/tools/exp_5d_profile.py $SHAPE > /tmp/prof_$TAG.log 2>&1
F=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: walk back from the end to the last absmax launch
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "absmax" in n)
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f us  grid %-10s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r["Kernel_Name"][:70]))
PY
