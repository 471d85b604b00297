#!/bin/bash
# two PMC passes (instruction mix, wait cycles) on the C2 bench (diagnostics; run via gpurun)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-launch-floor > $OUT/p$i.log 2>&1; echo "pass $i ($set): $?"
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep "k_interp\|k_spread" $OUT/summary.txt | cut -c1-300
