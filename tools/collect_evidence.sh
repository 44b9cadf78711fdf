#!/bin/bash
# Copies the results of a tools/round_evidence.sh run from gpurun_out/TAG into profiles/ under the
# names profiles/README.md lists. Usage: tools/collect_evidence.sh TAG [marker-file]
# (marker-file: only rocprofv3 CSVs newer than it are considered -- gpurun merges a run's files into
# what an earlier run with the same tag left behind).
TAG=$1; MARK=${2:-/dev/null}
O=gpurun_out/$TAG; P=profiles
newest_big() {  # largest file of the pattern that is newer than the marker
  find $1 -name "$2" -newer $MARK -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-
}
tail -1 $O/bench_with_traffic.log > $P/${TAG}_bench_line.json
tail -1 $O/bench_1024.log > $P/${TAG}_bench_line_1024.json
cp "$(newest_big $O/stats '*_kernel_stats.csv')" $P/${TAG}_kernel_stats_512cube_f32.csv
cp "$(newest_big $O/stats_step '*_kernel_stats.csv')" $P/${TAG}_kernel_stats_step_only_512cube_f32.csv
cp $O/pmc_raw.json $P/${TAG}_pmc_counters_512cube_f32.json
cp $O/pmc_raw_1024.json $P/${TAG}_pmc_counters_1024cube_f32.json
cp $O/sq_counters_chain.json $P/${TAG}_sq_counters_chain.json
cp $O/sq_counters_chain.txt $P/${TAG}_sq_counters_chain.txt
for c in 512:tl512 1024:tl1024 4d:tl4d f64:tlf64; do
  n=${c%%:*}; d=${c##*:}
  cp $O/$d/timeline.txt $P/${TAG}_timeline_$n.txt
  cp $O/$d/kernel_times.csv $P/${TAG}_step_only_kernel_times_$n.csv
done
cp $O/traffic_512cube_f32.json $O/traffic_1024cube_f32.json $O/traffic_512cube_f64nu.json $O/traffic_4d_slab_f32.json $P/
cp $O/pmc_raw_f64nu.json $P/${TAG}_pmc_counters_512cube_f64nu.json
cp $O/pmc_raw_4d.json $P/${TAG}_pmc_counters_4d_slab_f32.json
for f in hl_pipeline_lanes hl_pipeline_sequential hl_pipeline_ab recompose_profile 5d_profile 5d_trace e2e_512 grid_barrier host_link; do
  [ -f $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > $P/${TAG}_$f.txt
done
python - <<PY
import json, sys
sys.path.insert(0, '.')
import bench
h = bench.source_hash()
for f in ("traffic_512cube_f32.json", "traffic_1024cube_f32.json", "traffic_512cube_f64nu.json", "traffic_4d_slab_f32.json"):
    t = json.load(open("profiles/" + f))
    print(f, "source_hash", t.get("source_hash"), "matches" if t.get("source_hash") == h else "DOES NOT MATCH", h)
PY
cat $O/pytest_gpu.txt $O/smoke.txt
