#!/bin/bash
# Copies the summaries of tools/profile_round.sh runs from gpurun_out/ into profiles/ under this round's names.
R=$(cd "$(dirname "$0")/.." && pwd)
RND=${1:-r03}
for WL in C2 C3 C4 C5; do
  O=$R/gpurun_out/prof_round_$WL
  [ -f $O/summary.txt ] || continue
  wl=$(echo $WL | tr A-Z a-z)
  cp $O/summary.txt $R/profiles/${RND}_${wl}_summary.txt
  cp $O/summary.json $R/profiles/${RND}_${wl}_summary.json
  grep '^{' $O/bench.json | tail -1 > $R/profiles/${RND}_${wl}_bench.json
  f=$(ls -t $(find $O/stats -name "*kernel_stats.csv") 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/profiles/${RND}_${wl}_kernel_stats.csv
done
# instruction mix of the interpolation kernel (C2 passes), keyed by NUFFT width as bench.py reads it
python3 - $R $RND <<'PY'
import json, sys
R, RND = sys.argv[1:3]
try:
    s = json.load(open("%s/profiles/%s_c2_summary.json" % (R, RND)))
    mix = s.get("interp_instmix") or {}
    if mix:
        json.dump({"what": "k_interp_b dynamic instruction mix per wave from rocprofv3 --pmc passes on bench.py (BASELINE config 2, "
                           "two panels per launch), tools/profile_round.sh", "kernels_sha": s.get("kernels_sha"), "12": mix},
                  open("%s/profiles/%s_interp_instmix.json" % (R, RND), "w"), indent=1)
except Exception as e:
    print("no instmix:", e)
PY
ls -la $R/profiles | tail -20
