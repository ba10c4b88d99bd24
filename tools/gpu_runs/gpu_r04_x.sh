#!/bin/bash
# round 4: K1's pair loop, ingredient by ingredient (A: 64-bit trip count; B: + exchange by selects; C: + one table copy; D: + plain wrap; E: all)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
for round in 1 2; do
  for lib in libmaxigpu.so ab_old.so ab_A.so ab_B.so ab_C.so ab_D.so ab_E.so; do
    echo "## $lib round $round" | tee -a $O/ab.txt
    MXG_LIB=$R/maximilian_amd/$lib MODE=one REPS=300 timeout 300 python tools/sweep_heavy_osc.py 8 10 9 0 2>&1 | grep "^wf" | tr '\n' ' ' | tee -a $O/ab.txt; echo | tee -a $O/ab.txt
  done
done
