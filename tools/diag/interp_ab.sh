#!/bin/bash
# A/B of the k_interp generations on the GPU box: rocprofv3 kernel stats per variant and workload.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-interp_ab}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-1 2 3}; do
  for wl in C2 C3 C5; do
    SK_INTERP=$v python3 $R/tools/diag/devsteps.py $wl 50 >> $OUT/plain.txt 2>&1 || exit 1
  done
  SK_INTERP=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$v -- python3 $R/tools/diag/devsteps.py C2 50 > $OUT/v$v.log 2>&1 || exit 1
  f=$(find $OUT/v$v -name "*kernel_stats.csv" | head -1)
  echo "== variant $v" >> $OUT/stats.txt
  python3 - "$f" >> $OUT/stats.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %5s avg_us %8.2f pct %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
cat $OUT/plain.txt $OUT/stats.txt
