#!/bin/bash
# round 6, call k: K1t with one-round workgroups, two per CU (tab_sides 1) against the side-by-side pair (tab_sides 2), pipelined step, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q 2>&1 | tail -4 > $O/t_osctab.log
for r in 1 2 3; do for sd in 1 2; do
timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 --kernel-events pass --verbose --tune tab_sides=$sd 2>> $O/err.log | python tools/line_fields.py "tab_sides=$sd r$r"
done; done | tee $O/ab.txt
tail -n 3 $O/t_osctab.log
