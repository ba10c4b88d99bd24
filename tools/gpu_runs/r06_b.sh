#!/bin/bash
# round 6, call b: pipelined per-voice tables (K1t), the granular retry, the refused-call patch, config 3 mixdown at full size
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_osctab.py -x -q 2>&1 | tail -15 > $O/t_osctab.log
timeout 900 python -m pytest tests/test_gpu_grains.py -x -q 2>&1 | tail -15 > $O/t_grains.log
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -k "refused or granular" 2>&1 | tail -15 > $O/t_dropin.log
timeout 900 python -m pytest tests/test_gpu_fullparity.py -x -q -k "config3_mixdown or config5" -s 2>&1 | tail -15 > $O/t_full.log
for r in 1 2; do
timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 > $O/tables_new_$r.json 2>> $O/err.log
timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 --tune tables_serial=1 > $O/tables_old_$r.json 2>> $O/err.log
done
timeout 300 python bench.py --workload config5 --no-cpu-baseline --steps 20 --warmup 3 > $O/config5.json 2>> $O/err.log
for f in $O/t_*.log; do echo "== $f"; tail -n 4 $f; done
for f in $O/tables_*.json $O/config5.json; do python tools/line_fields.py $f < $f; done
