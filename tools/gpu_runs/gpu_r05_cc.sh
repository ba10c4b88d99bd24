# round 5, end: the one-launch granular calls on two streams at once (new test), the grain tests once more
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_grains.py -x -q -m gpu 2>&1 | tail -4
