#!/bin/bash
# round 4: the two-rank launches three times over (the ramp's step count is now agreed between the ranks)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in 1 2 3; do timeout 900 python -m pytest tests/test_bench_launch.py -q -x -k "two_ranks" 2>&1 | tail -1; done
