#!/bin/bash
# round 6: K2f plain mode A by bank size, with and without the instruction diet (interleaved)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06k2fk; mkdir -p $O
for V in 16384 32768 49152 65536 81920 98304 131072 262144; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_oldk2f.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --voices $V --no-cpu-baseline --no-extras --no-configs --steps 256 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeA V=$V"
done; done | tee $O/ab.txt
