#!/bin/bash
# round 4: K1's lean loop form (32-bit trip count, 4-instruction pair exchange, lean ticks) for EVERY waveform against the per-waveform choice
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04ao
mkdir -p $O
cd $R
for round in 1 2; do for lib in libmaxigpu.so ab_leanall.so; do
  echo "## $lib round $round" | tee -a $O/ab.txt
  MXG_LIB=$R/maximilian_amd/$lib MODE=one REPS=300 timeout 300 python tools/sweep_heavy_osc.py 2 3 4 5 6 7 8 11 2>&1 | grep "^wf" | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
done; done
