#!/bin/bash
# round 6: kernel trace of the N > 1 step on one GPU at 16 and 32 blocks per batch: where a batch's fixed cost sits (tools/trace_gaps.py)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06gaps; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 16 32; do
rocprofv3 --kernel-trace --output-format csv -d $O/m$m -o b -- python $R/bench.py --mixdown fused --mix-depth $m --no-cpu-baseline --kernel-events off --steps 640 --warmup 64 > $O/m$m.log 2>&1
python $R/tools/trace_gaps.py $O/m$m/b_kernel_trace.csv osc_mix $m > $O/gaps_m$m.txt 2>&1
done
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "b_kernel_trace.csv" -size +20M -delete
cat $O/gaps_m16.txt $O/gaps_m32.txt | cut -c1-400
