# round 5: K8d (maxiStretch / maxiPitchShift) as one launch too; config 5 with 256-lane scheduler workgroups and the flat mix fold
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_grains.py tests/test_gpu_fullparity.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q -k "grain or config5 or stretch or Grain or pitch or dropin or Stretch" 2>&1 | tail -5 | tee $O/tests.log
timeout 600 python tools/bench_grains_streamed.py 2>&1 | tail -4 | tee $O/k8d.log
for r in 1 2 3; do
  timeout 300 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python tools/line_fields.py "config5 r$r"
done | tee $O/bench.log
