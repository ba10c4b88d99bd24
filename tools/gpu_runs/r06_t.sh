#!/bin/bash
# round 6, call t: the ring-row sample bank under the time-part knob again (the over-fetch that made two parts best is gone)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06t; mkdir -p $O
for r in 1 2; do for t in 2 3 4 6 8; do
timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 60 --warmup 10 --kernel-events off --tune smp_split=$t 2>> $O/err.log | python tools/line_fields.py "ring smp_split=$t r$r"
done; done | tee $O/sweep.txt
