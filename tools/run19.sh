cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03p
timeout 900 python -m pytest tests/test_gpu_osc.py -m gpu -q -x -k "render_mix" > gpurun_out/r03p/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03p/tests.log
for var in 0 4; do for extra in "" "--mix-only"; do
  tag=v${var}$(echo $extra | tr -d ' -')
  timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --steps 200 --warmup 20 --tune osc_mix_var=$var $extra > gpurun_out/r03p/b_$tag.json 2> gpurun_out/r03p/b_$tag.err
done; done
timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --steps 200 --warmup 20 --tune osc_mix_var=4 --voices 131072 > gpurun_out/r03p/b_v4_131072.json 2> gpurun_out/r03p/b_v4_131072.err
timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --steps 200 --warmup 20 --tune osc_mix_var=0 --voices 131072 > gpurun_out/r03p/b_v0_131072.json 2> gpurun_out/r03p/b_v0_131072.err
