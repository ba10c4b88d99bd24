# round 5: K1t with the marks pass inside the render launch (knob tab_fused): parity, then the step under both settings, interleaved
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.log
for r in 1 2 3; do for f in 1 0; do
  timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 --tune tab_fused=$f 2>/dev/null | python tools/line_fields.py "tables tab_fused=$f r$r"
done; done | tee $O/bench.log
