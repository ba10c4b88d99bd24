#!/bin/bash
# round 6: kernel-trace timeline of one config-5 step on the final code (tools/trace_timeline.py)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c5tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o b -- python $R/bench.py --workload config5 --no-cpu-baseline --kernel-events off --steps 6 --warmup 3 > $O/log.txt 2>&1
python $R/tools/trace_timeline.py $O/kt/b_kernel_trace.csv granular_prologue 2 > $O/timeline.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
tail -30 $O/timeline.txt
