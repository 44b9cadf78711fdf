#!/bin/bash
# rocprofv3 kernel statistics of mgh_compress + mgh_decompress on ONE shape (tools/exp_one_shape.py):
# which kernels take the time on a shape outside the fused 3-D / 4-D path.
# Usage: tools/exp_shape_trace.sh 8192,8192 [float64]
SHAPE=${1:-8192,8192}; DT=${2:-float32}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_shape
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_shape -- python3 $GRAFT_REPO_ROOT/tools/exp_one_shape.py $SHAPE $DT 2>&1 | grep compress
F=$(find /tmp/prof_shape -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per compress + decompress: %.3f ms" % (tot / 3 / 1e6))
for r in rows[:14]:
    print("%8.1f us  x%-4d %5.1f %%  %s" % (float(r["TotalDurationNs"]) / 3 / 1e3, int(r["Calls"]) // 3, float(r["Percentage"]), r["Name"][:90]))
PY
