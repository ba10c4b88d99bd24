#!/bin/bash
# round 4: sinebuf4's rest case (phase == 0) behind a wave-level branch against a per-lane select of the value
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for r in 1 2 3; do for lib in libmaxigpu.so ab_sb4sel.so; do echo -n "$lib: "; MXG_LIB=$R/maximilian_amd/$lib MODE=one REPS=300 timeout 300 python tools/sweep_heavy_osc.py 9 2>&1 | grep "^wf" | tr '\n' ' '; echo; done; done
