#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_bench_launch.py tests/test_gpu_voice.py -q -x 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
