# round 5: both renderers enqueued = two complete one-launch forms (the one not picked returns at once); off-grid carried grains test
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_grains.py tests/test_gpu_fullparity.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q -k "grain or config5 or stretch or Grain or pitch or dropin or Stretch" 2>&1 | tail -5 | tee $O/tests.log
timeout 600 python tools/bench_grains_streamed.py 2>&1 | tail -4 | tee $O/k8d.log
for r in 1 2 3; do
  timeout 300 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python tools/line_fields.py "config5 r$r"
done | tee $O/bench.log
