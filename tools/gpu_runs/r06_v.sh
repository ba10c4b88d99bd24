#!/bin/bash
# round 6, call v: the ring sample kernel without its window loads (ablation): is the bank bound by load latency?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06v; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_smpnoload.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 60 --warmup 10 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "bank $lib r$r"
done; done | tee $O/ab.txt
