#!/bin/bash
# several values of one environment knob on another bench configuration:
# tools/ab_cfg.sh CONFIG VAR "v1 v2 v3" [bench args]
CFG=$1; V=$2; VALS=$3; shift 3
for x in $VALS; do
  env $V=$x python bench.py --config $CFG --only-step --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step_all']
print('$CFG $V=$x', d['ms_per_step'], {a:round(b*1000) for a,b in k.items()})"
done
