# round 5: the steps of the other multi-kernel workloads as timelines (is anything but the kernels in them?)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
tl() {  # name anchor periods bench-args...
  name=$1; anchor=$2; per=$3; shift 3
  rm -rf /tmp/t_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/t_$name -o t -- python $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 --kernel-events off "$@" > $O/$name.log 2>&1
  f=$(find /tmp/t_$name -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_timeline.py $f $anchor $per > $O/$name.timeline.txt 2>&1
  tail -14 $O/$name.timeline.txt
}
tl tables osctab_marks 2 --workload tables
tl mixdown osc_mix 3 --mixdown fused
tl config3mix voice_kernel 3 --workload config3 --mixdown fused
