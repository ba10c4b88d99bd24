#!/bin/bash
# round 6: play() on the controlled schedule: tests, then tools/bench_banks.py (automatic / never), twice
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace17; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sample.py tests/test_gpu_dropin.py -q -x 2>&1 | tail -3 > $O/t.log; tail -2 $O/t.log
for r in 1 2; do for p in 0 1; do
timeout 300 python tools/bench_banks.py smp_pace=$p 2>/dev/null | grep "^sample" | sed "s/^/smp_pace=$p /"
done; done | tee $O/ab.txt
