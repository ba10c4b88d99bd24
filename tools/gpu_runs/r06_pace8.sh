#!/bin/bash
# round 6: the headline kernel (K1, sinebuf, 65 536 voices) on the paced schedule: controller, fixed periods, store flavours
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06pace8; mkdir -p $O; rm -f $O/err.log

for r in 1 2; do for st in 0 2 5 4 3; do for p in 1 0 50 54 58; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --steps 512 --warmup 64 --kernel-events off --tune osc_pace=$p --tune osc_store=$st 2>> $O/err.log | python tools/line_fields.py "K1 store=$st osc_pace=$p r$r"
done; done; done | tee $O/ab.txt
grep "^pace" $O/err.log | tail -12
