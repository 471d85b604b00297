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
import csv, re, sys
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "sk::k_" in r["Name"] and "warmup" not in r["Name"] and int(r["Calls"]) >= 20:
        name = re.sub(r"^void ", "", r["Name"]).split("(")[0]  # full instantiation name: <W, K, waves> differ in cost
        rows[name] = "%.2f x%s" % (float(r["AverageNs"]) / 1e3, r["Calls"])
print(sys.argv[2], rows)
PY
done
