# round 5: K8c interior pairs: buffer[a] as an 8-byte load, buffer[a + 1] from the neighbour lane (wavefront shift) -- parity and config 5
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_grains.py tests/test_gpu_fullparity.py -m gpu -x -q -k "grain or config5" 2>&1 | tail -5 | tee $O/tests.log
for r in 1 2 3; do
  timeout 300 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python tools/line_fields.py "config5 r$r"
done | tee $O/bench.log
