#!/bin/bash
# round 6: K1's per-voice loads requested before the table is staged, against the build without (build/ab/ab_head.so), free-running kernel (osc_pace 1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pre; mkdir -p $O; rm -f $O/err.log
timeout 900 python -m pytest tests/test_gpu_osc.py -q -x 2>&1 | tail -1
for r in 1 2 3; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_head.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --steps 2000 --warmup 100 --no-configs --no-extras --no-cpu-baseline --kernel-events off --tune osc_pace=1 2>> $O/err.log | python tools/line_fields.py "$lib sinebuf r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --steps 2000 --warmup 100 --waveform saw --no-configs --no-extras --no-cpu-baseline --kernel-events off --tune osc_pace=1 2>> $O/err.log | python tools/line_fields.py "$lib saw r$r"
done; done | tee $O/ab.txt
