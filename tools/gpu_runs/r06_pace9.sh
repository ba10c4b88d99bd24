#!/bin/bash
# round 6: K1m (config 2's N > 1 step) on a paced schedule, fixed periods (ticks of 10 ns per 16 samples), bare kernel and with the mix queue
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace9; mkdir -p $O; rm -f $O/err.log
for r in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K1 r$r"
for p in 0 100 104 108 112 116 120; do
timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --no-configs --steps 512 --warmup 64 --kernel-events off --tune osc_mix_pace=$p 2>> $O/err.log | python tools/line_fields.py "K1m osc_mix_pace=$p r$r"
done; done | tee $O/ab.txt
