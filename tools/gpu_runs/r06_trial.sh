#!/bin/bash
# round 6: the headline with the on-device trial (free-running or paced) against the free-running kernel alone (osc_pace 1), long and short runs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06trial; mkdir -p $O; rm -f $O/err.log

for r in 1 2 3; do for p in 0 1; do
timeout 600 python bench.py --steps 2000 --warmup 100 --no-configs --no-extras --no-cpu-baseline --kernel-events off --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "2000 steps osc_pace=$p r$r"
timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-extras --no-cpu-baseline --kernel-events off --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "20 steps osc_pace=$p r$r"
done; done | tee $O/ab.txt
grep "^trial\|^pace" $O/err.log
