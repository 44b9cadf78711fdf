#!/bin/bash
# SQ / TCC counters + FETCH/WRITE traffic of the decompression side (dequantize + recompose leg of
# bench.py): PMC passes of the default legs without the other configurations.
# usage (through gpurun): tools/decomp_counters.sh TAG
TAG=$1; shift
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d $O/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline "$@" > $O/p$i.log 2>&1
  echo "pass $i rc=$?" >> $O/passes.txt
done
cd $R
python3 tools/chain_counters.py $O/decomp_counters.json $(ls $O/p*/*/*.db) > $O/decomp_counters.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/passes.txt; grep -i "loadvec\|restore\|head\|ipk\|dequant" $O/decomp_counters.txt
