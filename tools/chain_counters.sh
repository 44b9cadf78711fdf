#!/bin/bash
# SQ / TCC counters of every kernel of one step: four rocprofv3 PMC passes of
# `bench.py --only-step` (kernel-trace only, program directly after `--`), merged per kernel.
# usage (through gpurun): tools/chain_counters.sh TAG [bench args]
TAG=$1; shift
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
P4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d $O/p$i -- python3 $R/bench.py --only-step --steps 3 --warmup 1 "$@" > $O/p$i.log 2>&1
  echo "pass $i rc=$?" >> $O/passes.txt
done
cd $R
python3 tools/chain_counters.py $O/chain_counters.json $(ls $O/p*/*/*.db) > $O/chain_counters.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/passes.txt; cat $O/chain_counters.txt
