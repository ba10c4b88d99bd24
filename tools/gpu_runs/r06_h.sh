#!/bin/bash
# round 6, call h: the sample players' rows as rings (only new pieces fetched) against round 5's windows, HBM-resident bank and shared sample
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sample.py tests/test_gpu_rw_store.py -x -q 2>&1 | tail -5 > $O/t_sample.log
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_smpold.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 60 --warmup 10 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "bank $lib r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib REPS=20 timeout 300 python tools/bench_speedplayer.py 2>> $O/err.log | sed "s|^|shared $lib r$r |"
done; done | tee $O/ab.txt
tail -n 4 $O/t_sample.log
