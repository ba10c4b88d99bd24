#!/bin/bash
# round 6, call l: K1t tests on both workgroup forms; the tables step with the rows in the grouped mix queue (tables_queue 1) / with the row-sum kernel (0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_osctab.py tests/test_cabi.py -x -q 2>&1 | tail -4 > $O/t_osctab.log
for r in 1 2; do for q in 1 0; do
timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 208 --warmup 16 --kernel-events pass --verbose --tune tables_queue=$q 2>> $O/err.log | python tools/line_fields.py "tables_queue=$q r$r"
done; done | tee $O/ab.txt
tail -n 3 $O/t_osctab.log; tail -3 $O/err.log
