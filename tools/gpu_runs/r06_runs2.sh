#!/bin/bash
# round 6: K2f run loops with the stores in flight per wavefront bounded (s_waitcnt vmcnt(n) at the top of every steady chunk)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06runs2; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_noruns.so build/ab/ab_vm0.so build/ab/ab_vm8.so build/ab/ab_vm16.so build/ab/ab_vm32.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=2 2>> $O/err.log | python tools/line_fields.py "$lib modeA diet r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeA+mix r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeB r$r"
done; done | tee $O/ab.txt
