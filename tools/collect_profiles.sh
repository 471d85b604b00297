#!/bin/bash
# Copies the summaries of tools/profile_round.sh runs from gpurun_out/ into profiles/ under this round's names.
R=$(cd "$(dirname "$0")/.." && pwd)
RND=${1:-r02}
for WL in C2 C3 C4 C5; do
  O=$R/gpurun_out/prof_round_$WL
  [ -f $O/summary.txt ] || continue
  wl=$(echo $WL | tr A-Z a-z)
  cp $O/summary.txt $R/profiles/${RND}_${wl}_summary.txt
  cp $O/summary.json $R/profiles/${RND}_${wl}_summary.json
  grep '^{' $O/bench.json | tail -1 > $R/profiles/${RND}_${wl}_bench.json
  f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/profiles/${RND}_${wl}_kernel_stats.csv
done
ls -la $R/profiles | tail -20
