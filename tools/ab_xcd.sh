#!/bin/bash
# same-box A/B of the level kernel's tile placement (MGH_FUSED_XCD = 1 balanced ranges from 64 tiles on /
# 2 equal ranges from 32 tiles on, rounds 2-5 / 0 launch order): the metric's step, then other shapes.
for r in 1 2 3; do for v in 1 2 0; do
  echo -n "MGH_FUSED_XCD=$v  "; MGH_FUSED_XCD=$v python bench.py --only-step --steps 40 --warmup 5 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
