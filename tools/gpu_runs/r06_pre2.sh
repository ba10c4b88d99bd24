#!/bin/bash
# round 6: K1m's per-voice loads requested before the table and the gains are staged, against the build without (build/ab/ab_head.so)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pre2; mkdir -p $O; rm -f $O/err.log
timeout 900 python -m pytest tests/test_gpu_osc.py -q -x -k "mix" 2>&1 | tail -1
for r in 1 2 3; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_head.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --mixdown fused --steps 1000 --warmup 100 --no-configs --no-extras --no-cpu-baseline --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib config2 + mixdown r$r"
done; done | tee $O/ab.txt
