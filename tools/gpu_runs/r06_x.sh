#!/bin/bash
# round 6, call x: the sample suite on the final kernel (lean parts, two chunks in flight), the bank, the shared sample
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06x; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sample.py tests/test_gpu_rw_store.py tests/test_gpu_sampler.py -x -q 2>&1 | tail -4 > $O/t.log
for r in 1 2; do
timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 100 --warmup 20 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "bank r$r"
REPS=20 timeout 300 python tools/bench_speedplayer.py 2>> $O/err.log | sed "s|^|shared r$r |"
done | tee $O/ab.txt
tail -n 3 $O/t.log
