#!/bin/bash
# round 6, call r: K2f's steady paths with their pair-row stores spread over the chunk (A/B), config 3 mode A and B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06r; mkdir -p $O
for r in 1 2 3; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_vspread.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 256 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeA $lib r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 256 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeB $lib r$r"
done; done | tee $O/ab.txt
