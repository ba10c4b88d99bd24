#!/bin/bash
# round 4, second GPU call: parity of the passes loops + the sweep that decides K1 / K1m's grid
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_osc.py -x -q -m gpu -k "passes or render_mix or store_streams" > $O/pytest_osc.log 2>&1
echo "pytest rc=$?" >> $O/pytest_osc.log
tail -3 $O/pytest_osc.log
timeout 1500 python tools/sweep_osc_passes.py --out $O/osc_passes.md > $O/sweep.log 2>&1
grep -E "^\*\*" $O/osc_passes.md | head -20
