#!/bin/bash
# Round evidence on one GPU box: GPU tests, smoke, bench line, rocprofv3 kernel stats of the bench
# command, step timelines, PMC traffic passes (512^3 and 1024^3). Usage (through gpurun):
#   tools/round_evidence.sh r03z
TAG=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "^FAILED|^ERROR|passed|failed" | tail -12 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-other-configs > $O/stats.log 2>&1
# the same with the timed steps only (no decompress / sym16 / two-stream / end-to-end legs, whose
# launches of the same kernels -- some of them overlapped on two streams -- enter the averages above)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -- python3 $R/bench.py --steps 10 --warmup 3 --only-step > $O/stats_step.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --only-step > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --only-step > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch1024 -- python3 $R/bench.py --config 1024f32 --steps 3 --warmup 1 --only-step > $O/fetch1024.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write1024 -- python3 $R/bench.py --config 1024f32 --steps 3 --warmup 1 --only-step > $O/write1024.log 2>&1
for c in 512f64nu:f64nu 4d:4d; do
  cfg=${c%%:*}; d=${c##*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$d -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --only-step > $O/fetch_$d.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$d -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --only-step > $O/write_$d.log 2>&1
done
# the two-lane subdomain pipeline of mgh_compress / mgh_decompress on the 64 x 512^3 volume: which
# queue runs what, pipelined and sequential (MGH_HL_PIPELINE=0)
rocprofv3 --kernel-trace -d $O/lanes1 -- python3 $R/tools/exp_hl_pipeline.py 64 1 --once > $O/lanes1.log 2>&1
rocprofv3 --kernel-trace -d $O/lanes0 -- python3 $R/tools/exp_hl_pipeline.py 64 1 --once --seq > $O/lanes0.log 2>&1
cd $R
python3 tools/lanes_timeline.py $(ls $O/lanes1/*/*.db | head -1) --window-ms 170 --min-us 250 > $O/hl_pipeline_lanes.txt 2>&1
python3 tools/lanes_timeline.py $(ls $O/lanes0/*/*.db | head -1) --window-ms 185 --min-us 250 > $O/hl_pipeline_sequential.txt 2>&1
python3 tools/exp_hl_pipeline.py 64 3 > $O/hl_pipeline_ab.txt 2>&1
python3 tools/exp_recompose_profile.py > $O/recompose_profile.txt 2>&1
python3 tools/exp_5d_profile.py > $O/5d_profile.txt 2>&1
python3 tools/exp_e2e_out.py > $O/e2e_512.txt 2>&1
[ -x tools/micro/grid_barrier ] && ./tools/micro/grid_barrier > $O/grid_barrier.txt 2>&1
[ -x tools/micro/host_link ] && ./tools/micro/host_link 512 > $O/host_link.txt 2>&1
bash tools/exp_5d_trace.sh ${TAG}_5d > $O/5d_trace.txt 2>&1
rm -rf $O/lanes1 $O/lanes0
python tools/make_traffic.py $(ls $O/fetch_f64nu/*/*.db | head -1) $(ls $O/write_f64nu/*/*.db | head -1) $O/pmc_raw_f64nu.json traffic_512cube_f64nu.json > $O/traffic_f64nu.txt 2>&1
python tools/make_traffic.py $(ls $O/fetch_4d/*/*.db | head -1) $(ls $O/write_4d/*/*.db | head -1) $O/pmc_raw_4d.json traffic_4d_slab_f32.json > $O/traffic_4d.txt 2>&1
rm -rf $O/fetch_f64nu $O/write_f64nu $O/fetch_4d $O/write_4d
python tools/make_traffic.py $(ls $O/fetch/*/*.db | head -1) $(ls $O/write/*/*.db | head -1) $O/pmc_raw.json > $O/traffic.txt 2>&1
python tools/make_traffic.py $(ls $O/fetch1024/*/*.db | head -1) $(ls $O/write1024/*/*.db | head -1) $O/pmc_raw_1024.json traffic_1024cube_f32.json > $O/traffic1024.txt 2>&1
cp profiles/traffic_512cube_f32.json profiles/traffic_1024cube_f32.json profiles/traffic_512cube_f64nu.json profiles/traffic_4d_slab_f32.json $O/
# timelines of one step (kernel trace, no counters)
tools/trace_step.sh $TAG/tl512
tools/trace_step.sh $TAG/tl1024 --config 1024f32
tools/trace_step.sh $TAG/tl4d --config 4d
tools/trace_step.sh $TAG/tlf64 --config 512f64nu
# SQ / TCC counters of every kernel of the step (final code)
tools/chain_counters.sh $TAG/chain > /dev/null 2>&1
cp $O/chain/chain_counters.json $O/sq_counters_chain.json 2>/dev/null
cp $O/chain/chain_counters.txt $O/sq_counters_chain.txt 2>/dev/null
python bench.py > $O/bench_with_traffic.log 2>&1
python bench.py --config 1024f32 --only-step > $O/bench_1024.log 2>&1
rm -rf $O/fetch $O/write $O/fetch1024 $O/write1024
cat $O/pytest_gpu.txt $O/smoke.txt; tail -1 $O/bench_with_traffic.log | cut -c1-400
