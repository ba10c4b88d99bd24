#!/bin/bash
# round 6: the driver-style 20-step headline run under the runtime's signal-wait policies (wall clock carries the final synchronize's wake-up)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06sync; mkdir -p $O
for r in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --kernel-events off 2>> $O/err.log | python tools/line_fields.py "default r$r"
HSA_ENABLE_INTERRUPT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --kernel-events off 2>> $O/err.log | python tools/line_fields.py "HSA_ENABLE_INTERRUPT=0 r$r"
done | tee $O/ab.txt
