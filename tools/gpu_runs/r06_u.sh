#!/bin/bash
# round 6, call u: the ring sample kernel with two chunks in flight: parity + time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sample.py -x -q -k "ring_rows or speed_players or time_parts" 2>&1 | tail -4 > $O/t.log
for r in 1 2 3; do
timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 60 --warmup 10 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "bank r$r"
done | tee $O/ab.txt
tail -n 3 $O/t.log
