#!/bin/bash
# Step time and per-kernel times under developer switches, same box: tools/sweep_env.sh (through gpurun)
run() { env "$@" python bench.py --only-step --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step_all']
print('%-34s' % ' '.join(sys.argv[1:]), d['ms_per_step'], {a:round(b*1000) for a,b in k.items()})" "$@"; }
for i in 1 2; do
run A=0
run MGH_RCH=1,3,16
run MGH_RCH=1,5,16
run MGH_RCH=2,4,16
run MGH_RCH=1,8,16
run MGH_CLS2=1024
run MGH_CLS2=512 MGH_RCH=1,4,8
run MGH_CLS1=512
run MGH_BOX=2
run MGH_FUSED_WIDE=2
run MGH_FUSED_WIDE=0
done
