cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_gpu_grains.py -m gpu -x -q 2>&1 | tail -2
for X in "-DMXG_UNIT_BATCH=8" "-DMXG_UNIT_BATCH=12"; do
  touch maximilian_amd/csrc/grains.hip; make -C maximilian_amd/csrc EXTRA="$X" >/dev/null 2>&1
  echo "EXTRA=$X"
  timeout 100 bash tools/profile_configs.sh x grains
  grep -E "unit_kernel" gpurun_out/prof_x_configs/grains/k_kernel_stats.csv | cut -d, -f1,4 | cut -c30-140
done
