# round 5: config 5's step as a timeline (which kernel waits for which): rocprofv3 kernel trace, no event pairs inside the step
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/t5 -o t5 -- python $R/bench.py --workload config5 --no-cpu-baseline --steps 6 --warmup 3 --kernel-events off > $O/bench.log 2>&1
f=$(find /tmp/t5 -name "*kernel_trace.csv" | head -1); tail -80 $f | cut -c1-300 > $O/trace_tail.csv
python $R/tools/trace_timeline.py $f granular_unit_check 2 | tee $O/timeline.txt | tail -60
