#!/bin/bash
# rocprofv3 kernel stats of a python script: prof.sh <outdir under gpurun_out> <script> [args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/"$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/stats.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print("%-70s calls %5s avg_us %9.2f total_ms %8.2f pct %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
cat $OUT/run.log | tail -4; cat $OUT/stats.txt
