#!/bin/bash
# rocprofv3 kernel trace of `bench.py --only-step` + the timeline of its last step.
# usage (through gpurun): tools/trace_step.sh TAG [bench args]   (environment switches are inherited)
TAG=$1; shift
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -- python3 $R/bench.py --only-step --steps 6 --warmup 2 "$@" > $O/trace.log 2>&1
cd $R
DB=$(ls $O/trace/*/*.db | head -1)
python3 tools/timeline.py $DB > $O/timeline.txt 2>&1
python3 tools/kernel_times.py $DB --csv $O/kernel_times.csv > /dev/null 2>&1
rm -rf $O/trace
