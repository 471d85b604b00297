#!/bin/bash
# Diagnostic: spread tile geometry (SK_TILE_CELLS / SK_TILE_SRC knobs of build_tiles, sk_api.cpp) against ms per step.
# usage: tile_sweep.sh <workload> "<cells list>" "<src list>"
wl=${1:-C5}
for c in ${2:-64 32 16}; do for s in ${3:-256 128}; do
  echo "cells=$c src=$s: $(SK_TILE_CELLS=$c SK_TILE_SRC=$s python tools/diag/devsteps.py $wl 200 2>&1 | tail -1 | cut -c1-260)"
done; done
