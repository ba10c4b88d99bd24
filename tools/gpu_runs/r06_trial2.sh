#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06trial2; mkdir -p $O; rm -f $O/err.log
for r in 1 2; do for p in 54 58 62; do
timeout 600 python bench.py --steps 512 --warmup 64 --no-configs --no-extras --no-cpu-baseline --kernel-events off --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "pair-row kernel, paced branch, P=$p r$r"
timeout 600 python bench.py --steps 512 --warmup 64 --no-configs --no-extras --no-cpu-baseline --kernel-events off --tune osc_pace=$p --tune osc_vpl=1 --tune osc_store=2 --tune osc_split=1 --tune osc_passes=1 --tune osc_plan=1 2>> $O/err.log | python tools/line_fields.py "8-byte kernel, paced, P=$p r$r"
done; done | tee $O/ab.txt
