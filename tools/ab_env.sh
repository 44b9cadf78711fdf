#!/bin/bash
# A/B of one environment knob on one box: tools/ab_env.sh VAR valueA valueB [bench args]
V=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do
  for x in $A $B; do
    env $V=$x python bench.py --only-step --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step_all']
print('$V=$x', d['ms_per_step'], {a:round(b*1000) for a,b in k.items()})"
  done
done
