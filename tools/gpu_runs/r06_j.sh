#!/bin/bash
# round 6, call j: K1t's render kernel taken apart (no rendering / no DMA in the loop), pipelined step, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_tab1.so build/ab/ab_tab2.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 --kernel-events pass --verbose 2>> $O/err.log | python tools/line_fields.py "$lib r$r"
done; done | tee $O/ab.txt
