#!/bin/bash
# round 6: K2f run loops with a pause per chunk (s_sleep n = 64 n cycles): does the store-bound size want a slower cadence?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06runs3; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_pace2.so build/ab/ab_pace4.so build/ab/ab_pace6.so build/ab/ab_pace8.so build/ab/ab_pace12.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=2 2>> $O/err.log | python tools/line_fields.py "$lib modeA diet r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=2 --tune voice_store=4 2>> $O/err.log | python tools/line_fields.py "$lib modeA diet pair sc1 r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeA+mix r$r"
done; done | tee $O/ab.txt
