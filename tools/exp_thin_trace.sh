#!/bin/bash
# rocprofv3 kernel trace of tools/exp_thin.py on ONE shape: every launch of the last step with grid
# and duration. Usage: tools/exp_thin_trace.sh 16395,39,39:f64
SPEC=${1:-16395,39,39:f64}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_thin
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_thin -- python3 $GRAFT_REPO_ROOT/tools/exp_thin.py $SPEC > /tmp/prof_thin.log 2>&1
F=$(find /tmp/prof_thin -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "absmax" in n or "sqsum" in n)
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f us  grid %-9s wg %-4s lds %-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r["Kernel_Name"][:80]))
PY
