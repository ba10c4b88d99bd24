# round 5: config 5 as ONE launch (scheduler lanes beside the tile renders): parity, step time, timeline
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_grains.py tests/test_gpu_fullparity.py tests/test_gpu_dropin.py -m gpu -x -q -k "grain or config5 or stretch or Grain or pitch or dropin" 2>&1 | tail -5 | tee $O/tests.log
for r in 1 2 3; do
  timeout 300 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python tools/line_fields.py "config5 r$r"
done | tee $O/bench.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/t5 -o t5 -- python $R/bench.py --workload config5 --no-cpu-baseline --steps 6 --warmup 3 --kernel-events off > $O/trace_bench.log 2>&1
f=$(find /tmp/t5 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py $f granular_prologue 1 | tee $O/timeline.txt | tail -30
cd $R
for r in 1 2; do
  timeout 300 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 --tune grain_streamed=0 2>/dev/null | python tools/line_fields.py "config5 sliced r$r"
done | tee -a $O/bench.log
