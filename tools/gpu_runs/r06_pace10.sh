#!/bin/bash
# round 6: K2f paced at other bank sizes (fixed periods around V / 65 536 x 56 ticks)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace10; mkdir -p $O; rm -f $O/err.log
for V in 49152 81920 98304 131072 196608; do
base=$(( V * 56 / 65536 ))
for f in 0 90 95 100 105 110 120; do
if [ $f = 0 ]; then p=1; else p=$(( base * f / 100 )); fi
timeout 300 python bench.py --workload config3 --voices $V --no-cpu-baseline --no-extras --no-configs --steps 256 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "modeA V=$V pace=$p"
done; done | tee $O/ab.txt
