cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $GRAFT_REPO_ROOT/tools/exp_e2e.py 2>&1 | grep -v amdgpu | head -1
timeout 300 python3 $GRAFT_REPO_ROOT/tools/exp_e2e.py 2>&1 | grep -v amdgpu | head -1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python3 $GRAFT_REPO_ROOT/tools/exp_e2e.py > /dev/null 2>&1; f=$(find /tmp/prof_x -name "*kernel_stats.csv" | head -1); python3 $GRAFT_REPO_ROOT/tools/csv_kernels.py $f | head -14
