#!/bin/bash
run() { env "$@" python bench.py --only-step --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step_all']
print('%-34s' % ' '.join(sys.argv[1:]), d['ms_per_step'], {a:round(b*1000) for a,b in k.items()})" "$@"; }
for i in 1 2; do
run A=0
run MGH_IPK_CONTIG=2
run MGH_ABSMAX_WARM_MB=128
run MGH_ABSMAX_WARM_MB=256
run MGH_BOX=2
run MGH_RCH=1,6,16
run MGH_CLS1=100000
run MGH_IPK_WPC=5
done
