cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03t
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_fullparity.py -m gpu -q -x -k "not config4" > gpurun_out/r03t/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03t/tests.log
for wf in sinebuf4 sinebuf; do
  timeout 300 python bench.py --waveform $wf --no-cpu-baseline --no-extras --steps 300 --warmup 30 > gpurun_out/r03t/b_$wf.json 2> gpurun_out/r03t/b_$wf.err
done
timeout 300 python bench.py --waveform sinebuf4 --mixdown fused --no-cpu-baseline --no-extras --steps 300 --warmup 30 > gpurun_out/r03t/b_sinebuf4_mix.json 2> gpurun_out/r03t/b_sinebuf4_mix.err
