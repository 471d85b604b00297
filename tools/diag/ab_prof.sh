#!/bin/bash
# same-box rocprofv3 kernel averages of two builds: ab_prof.sh <other .so relative to the repo> [workload]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=$R/$1
WL=${2:-C2}
OUT=$R/gpurun_out/ab_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in other new other new; do
  d=$OUT/${v}_$RANDOM
  if [ $v = other ]; then export SPECTRALKERNELS_HIP_LIB=$OTHER; else unset SPECTRALKERNELS_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/diag/devsteps.py $WL 200 > $d.log 2>&1 || exit 1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys
rows = {r["Name"][:24]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
print(sys.argv[2], {k: round(v, 2) for k, v in rows.items() if "sk::k_" in k and "warmup" not in k})
PY
done
