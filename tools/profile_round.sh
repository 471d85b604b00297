#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats and PMC passes of the same command for one
# workload; summaries land in gpurun_out/prof_round_<WL>/ and are copied into profiles/ (tools/collect_profiles.sh).
# PMC counters are collected in passes of their own with --kernel-trace only (no other trace domains).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-C2}
MODE=${2:-full}   # full: bench line + kernel stats + PMC passes; stats: bench line + kernel stats only
OUT=$R/gpurun_out/prof_round_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload $WL > $OUT/bench.json 2> $OUT/bench.err; echo "bench $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-host-leg --no-launch-floor > $OUT/stats.log 2>&1; echo "stats $?"
if [ "$MODE" = full ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-host-leg --no-launch-floor > $OUT/pmc_fetch.log 2>&1; echo "pmc_fetch $?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-host-leg --no-launch-floor > $OUT/pmc_write.log 2>&1; echo "pmc_write $?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-host-leg --no-launch-floor > $OUT/pmc_sq.log 2>&1; echo "pmc_sq $?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $OUT/pmc_f64 -- python3 $R/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-host-leg --no-launch-floor > $OUT/pmc_f64.log 2>&1; echo "pmc_f64 $?"
fi
python3 $R/tools/summarise_profiles.py $OUT > $OUT/summary.txt 2>&1; echo "summary $?"
