#!/bin/bash
# round 4, eighth GPU call: the per-voice wavetable extension (parity, first timing); the K1 launch plan at large sizes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q -m gpu > $O/pytest_osctab.log 2>&1
echo "pytest rc=$?" >> $O/pytest_osctab.log
tail -15 $O/pytest_osctab.log
timeout 600 python tools/sweep_osc_plan.py > $O/osc_plan.md 2>&1
cat $O/osc_plan.md
timeout 300 python tools/bench_osctab.py > $O/osctab.txt 2>&1
cat $O/osctab.txt
