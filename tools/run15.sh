cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03o
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullparity.py -m gpu -q -x -k "not trig and not large" > gpurun_out/r03o/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03o/tests.log
for lay in 1 2; do for ex in 1 0; do
  timeout 300 python bench.py --workload config4 --no-cpu-baseline --no-extras --steps 20 --warmup 5 --tune fused_layout=$lay --tune fft_exact=$ex > gpurun_out/r03o/b_${lay}_${ex}.json 2> gpurun_out/r03o/b_${lay}_${ex}.err
done; done

