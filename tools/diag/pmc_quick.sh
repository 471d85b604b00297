#!/bin/bash
# two PMC passes (instruction counts, FP64 mix) on the C2 bench: pmc_quick.sh <outdir under gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_quick2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAIT_INST_ANY" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-launch-floor > $OUT/p$i.log 2>&1; echo "pass $i ($set): $?"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cut -c1-400 $OUT/summary.txt
