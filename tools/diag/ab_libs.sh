#!/bin/bash
# same-box A/B of two builds of the library: alternating device-resident step times (tools/diag/devsteps.py)
# usage: ab_libs.sh <other .so relative to the repo> [workloads...]   (the in-tree library is "new")
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=$R/$1; shift
WLS=${@:-C2 C3 C5}
OUT=$R/gpurun_out/ab_libs.txt
: > $OUT
run_other() { SPECTRALKERNELS_HIP_LIB=$OTHER python3 $R/tools/diag/devsteps.py $1 300 2>&1 | sed 's/^/other: /' >> $OUT; }
run_new() { python3 $R/tools/diag/devsteps.py $1 300 2>&1 | sed 's/^/new:   /' >> $OUT; }
for i in 1 2; do  # ABBA: the second process of a pair measured ~1.5 % faster whatever it ran
  for wl in $WLS; do
    run_other $wl && run_new $wl && run_new $wl && run_other $wl || exit 1
  done
done
cut -c1-60 $OUT | sort -k2,2 -k1,1
