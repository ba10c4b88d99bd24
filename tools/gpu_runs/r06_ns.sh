#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06ns; mkdir -p $O; rm -f $O/err.log
for r in 1 2 3; do for p in 0 1; do for n in "48 50" "64 480"; do set -- $n
timeout 300 python bench.py --voices 131072 --no-cpu-baseline --no-extras --no-configs --steps $2 --warmup $1 --kernel-events off --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "sinebuf 131072 osc_pace=$p warm $1 steps $2 r$r"
done; done; done | tee $O/ab.txt
grep "^pace" $O/err.log
