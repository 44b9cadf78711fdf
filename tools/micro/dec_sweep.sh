cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python3 $GRAFT_REPO_ROOT/tools/exp_hl_timeline.py > /dev/null 2>&1; f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1); echo "$1: $(python3 $GRAFT_REPO_ROOT/tools/csv_kernels.py $f decode)"; }
for tb in 11 12 13; do export MGH_HUFF_TB=$tb; run tb$tb; done
