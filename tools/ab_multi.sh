#!/bin/bash
# several values of one environment knob, two rounds: tools/ab_multi.sh VAR "v1 v2 v3" [bench args]
V=$1; VALS=$2; shift 2
for i in 1 2; do
  for x in $VALS; do
    env $V=$x python bench.py --only-step --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step_all']
print('$V=$x', d['ms_per_step'], {a:round(b*1000) for a,b in k.items()})"
  done
done
