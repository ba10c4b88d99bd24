#!/bin/bash
# round 6, call c: K1t step forms A/B on one box (ahead x in-kernel sum), tests again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q 2>&1 | tail -15 > $O/t_osctab.log
timeout 900 python -m pytest tests/test_gpu_fullparity.py -x -q -k "config3_mixdown" -s 2>&1 | tail -15 > $O/t_full.log
for r in 1 2; do
for a in 0 1; do for m in 0 1; do
timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 --kernel-events off --tune tables_ahead=$a --tune tables_sum=$m 2>> $O/err.log | python tools/line_fields.py "ahead=$a sum=$m r$r"
done; done; done
for f in $O/t_*.log; do echo "== $f"; tail -n 4 $f; done
