#!/bin/bash
# Does the placement of the process (NUMA node of its threads / first-touch memory) matter for the
# host-buffer path? GPU's node against the other one, copy threads pinned next to the GPU
# (MGH_HL_COPY_AFFINITY=1, default) or left to the scheduler (=0). Dev tool (through gpurun).
python3 - <<'PY'
import glob
for d in glob.glob("/sys/class/drm/card*/device"):
    try:
        if open(d + "/vendor").read().strip() != "0x1002": continue
        print(d, "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip())
    except OSError:
        pass
PY
for aff in 1 0; do
for cpus in 0-63 64-127; do
  echo "== MGH_HL_COPY_AFFINITY=$aff taskset -c $cpus"
  for i in 1 2 3; do MGH_HL_COPY_AFFINITY=$aff taskset -c $cpus python3 tools/exp_host_sweep.py 2>&1 | tail -1; done
done
done
