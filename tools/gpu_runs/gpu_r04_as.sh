#!/bin/bash
# round 4: maxiFilter pair-row kernel without per-sample / per-pair scalar tests in its whole chunks -- parity, then A/B against the form with them
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_voice.py tests/test_gpu_edges.py -q -x 2>&1 | tail -2
bash tools/gpu_runs/gpu_r04_al.sh
