#!/bin/bash
# Round evidence on one GPU box: GPU tests, smoke, bench line, rocprofv3 kernel stats of the bench
# command, PMC traffic passes. Usage (through gpurun): tools/round_evidence.sh r02b
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py > $O/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 3 > $O/stats.log 2>&1
# the same with the timed steps only (no decompress / sym16 / two-stream / end-to-end legs, whose
# launches of the same kernels -- some of them overlapped on two streams -- enter the averages above)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -- python3 $R/bench.py --steps 10 --warmup 3 --only-step > $O/stats_step.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --only-step > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --only-step > $O/write.log 2>&1
cd $R
python tools/make_traffic.py $(ls $O/fetch/*/*.db | head -1) $(ls $O/write/*/*.db | head -1) $O/pmc_raw.json > $O/traffic.txt 2>&1
cp profiles/traffic_512cube_f32.json $O/traffic_512cube_f32.json
python bench.py > $O/bench_with_traffic.log 2>&1
cat $O/pytest_gpu.txt $O/smoke.txt; tail -1 $O/bench_with_traffic.log | cut -c1-400
ls $O/stats/*/ | head
