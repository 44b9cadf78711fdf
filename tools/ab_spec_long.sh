#!/bin/bash
# same-box A/B of MGH_IPK_SPEC_LONG (strided pencils of that length and more in verified chunks; 0: never)
for v in 0 2048 1024 512 256; do echo "MGH_IPK_SPEC_LONG=$v"; MGH_IPK_SPEC_LONG=$v python tools/exp_thin.py 16395,64,64:f32 16395,39,39:f64 8,16395,39,39:f64 2048,129,129:f64 4096,100,100:f32 1024,1024,1024:f32 2>&1 | grep "GB/s"; done
