#!/bin/bash
# round 6: which part of K2f's instruction diet costs plain mode A its 2.5 us -- one switch off per build, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06k2fj; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_oldk2f.so build/ab/ab_nosaw.so build/ab/ab_nospec.so build/ab/ab_nocarry.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeA r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_store=4 2>> $O/err.log | python tools/line_fields.py "$lib modeA pair rows nt r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeA+mix r$r"
done; done | tee $O/ab.txt
