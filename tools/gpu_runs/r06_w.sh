#!/bin/bash
# round 6, call w: the lean two-chunks-in-flight parts of the ring sample kernel against the committed kernel, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06w; mkdir -p $O
for r in 1 2 3; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_smphead.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 60 --warmup 10 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "bank $lib r$r"
done; done | tee $O/ab.txt
