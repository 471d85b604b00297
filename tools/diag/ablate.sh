#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for A in 2 4; do
  SPECTRALKERNELS_HIP_LIB=$R/tools/diag/libsk_abl$A.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl$A -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/abl$A.log 2>&1
  echo "ABL $A: $(grep k_interp $R/gpurun_out/abl$A/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200)"
done
