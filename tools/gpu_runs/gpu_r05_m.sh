# round 5: the whole GPU suite again (bench-launch tests updated: config3_modB, config4_walk, per-config cpu baselines, sharded config5 at N = 2)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
timeout 3300 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
