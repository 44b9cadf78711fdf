#!/bin/bash
# A/B/C... of several builds of the library on one box: tools/ab_n.sh "libA.so libB.so ..." [bench args]
LIBS=$1; shift
for i in 1 2 3; do
  for L in $LIBS; do
    MGARD_HIP_LIB=$PWD/$L python bench.py --only-step --steps 30 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step_all']
print('$L', d['ms_per_step'], {a:round(b*1000) for a,b in k.items() if a.startswith('level_fused')})"
  done
done
