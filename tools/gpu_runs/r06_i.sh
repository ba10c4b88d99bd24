#!/bin/bash
# round 6, call i: ring rows chosen by sample size -- tests, the two banks again, bench_banks table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sample.py tests/test_cabi.py -x -q 2>&1 | tail -5 > $O/t_sample.log
for r in 1 2; do
timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 60 --warmup 10 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "bank r$r"
REPS=20 timeout 300 python tools/bench_speedplayer.py 2>> $O/err.log | sed "s|^|shared r$r |"
done | tee $O/ab.txt
timeout 1200 python tools/bench_banks.py > $O/banks.txt 2>> $O/err.log
tail -n 4 $O/t_sample.log; tail -n 40 $O/banks.txt
