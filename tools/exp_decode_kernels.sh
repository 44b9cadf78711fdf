#!/bin/bash
# Kernel time of the Huffman decoder variants on the benchmark's record (rocprofv3 --kernel-trace --stats of
# tools/exp_decode_diag.py): MGH_HUFF_LEAN / MGH_HUFF_PAIR select the kernel. Usage: tools/exp_decode_kernels.sh [TOL]
TOL=${1:-1e-3}
PROG=${PROG:-"tools/exp_decode_diag.py 512,512,512 $TOL"}   # (int64 output; PROG=tools/exp_e2e_out.py: mgh_decompress, 16-bit symbols)
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-"MGH_HUFF_LEAN=1" "MGH_HUFF_LEAN=0" "MGH_HUFF_PAIR=2"}; do
  rm -rf /tmp/prof_dec
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python3 $GRAFT_REPO_ROOT/$PROG > /tmp/prof_dec.log 2>&1
  F=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_decode" in r["Name"]:
        print("   %-40s calls %s  avg %.1f us  min %.1f us" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
