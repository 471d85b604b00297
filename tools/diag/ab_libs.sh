#!/bin/bash
# same-box A/B of two builds of the library: alternating device-resident step times (tools/diag/devsteps.py)
# usage: ab_libs.sh <other .so relative to the repo> [workloads...]   (the in-tree library is "new")
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=$R/$1; shift
WLS=${@:-C2 C3 C5}
OUT=$R/gpurun_out/ab_libs.txt
: > $OUT
for i in 1 2 3; do
  for wl in $WLS; do
    SPECTRALKERNELS_HIP_LIB=$OTHER python3 $R/tools/diag/devsteps.py $wl 300 2>&1 | sed 's/^/other: /' >> $OUT || exit 1
    python3 $R/tools/diag/devsteps.py $wl 300 2>&1 | sed 's/^/new:   /' >> $OUT || exit 1
  done
done
cut -c1-60 $OUT | sort -k2,2 -k1,1
