cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_voice.py tests/test_gpu_osc.py -q -x -k "pair_row or noise" 2>&1 | tail -3
