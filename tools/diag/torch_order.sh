#!/bin/bash
# device-resident step time with the library's own HIP runtime (/opt/rocm) against torch's bundled one (torch imported
# first, as bench.py does at N > 1): alternating runs, no profiler
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/torch_order.txt
: > $OUT
for i in 1 2 3; do
  for wl in C2 C5; do
    python3 $R/tools/diag/devsteps.py $wl 200 >> $OUT 2>&1 || exit 1
    SK_TORCH_FIRST=1 python3 $R/tools/diag/devsteps.py $wl 200 2>&1 | sed 's/^/torch-first: /' >> $OUT || exit 1
  done
done
cut -c1-150 $OUT
