#!/bin/bash
# round 6: the two-stage K2f, where its time goes: ablated builds (back without filter / stores, front without envelope / oscillator, no priority)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06split2; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_split1.so build/ab/ab_split2.so build/ab/ab_split4.so; do
for vm in 0 1; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --voice-mode $vm --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_split=2 2>> $O/err.log | python tools/line_fields.py "$lib mode $vm split r$r"
done; done; done | tee $O/ab.txt
