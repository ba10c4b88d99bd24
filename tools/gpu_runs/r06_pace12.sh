#!/bin/bash
# round 6: K1 (sinebuf) at other bank sizes on a fixed-period schedule, one voice per lane, one pass, no launch plan; against its own automatic launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace12; mkdir -p $O; rm -f $O/err.log
for V in 49152 73728 98304 131072 196608; do
timeout 300 python bench.py --voices $V --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K1 V=$V automatic"
base=$(( V * 56 / 65536 ))
for st in 2 4; do for f in 0 95 100 105 110; do
if [ $f = 0 ]; then p=1; else p=$(( base * f / 100 )); fi
timeout 300 python bench.py --voices $V --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off --tune osc_vpl=1 --tune osc_plan=1 --tune osc_passes=1 --tune osc_split=1 --tune osc_store=$st --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "K1 V=$V 1v store=$st pace=$p"
done; done; done | tee $O/ab.txt
