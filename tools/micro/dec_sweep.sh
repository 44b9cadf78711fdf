cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -- python3 $GRAFT_REPO_ROOT/tools/exp_hl_timeline.py > /dev/null 2>&1; f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1); echo "$1: $(python3 $GRAFT_REPO_ROOT/tools/csv_kernels.py $f encode)"; }
run full
export MGARD_HIP_LIB=$GRAFT_REPO_ROOT/tools/micro/lib_dbgE1.so; run staging_only
export MGARD_HIP_LIB=$GRAFT_REPO_ROOT/tools/micro/lib_dbgE2.so; run upto_lookback
export MGARD_HIP_LIB=$GRAFT_REPO_ROOT/tools/micro/lib_dbgE3.so; run no_lookback
