#!/bin/bash
# round 6, call f: the HBM-resident sample bank under the time-part / pipeline / store knobs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
for t in "smp_split=0" "smp_split=1" "smp_split=2" "smp_split=3" "smp_split=4" "smp_split=8" "smp_pipe=0" "rw_store=1" "rw_store=3" "rw_store=4"; do
timeout 300 python bench.py --workload sample_bank --no-cpu-baseline --steps 60 --warmup 10 --kernel-events off --tune $t 2>> $O/err.log | python tools/line_fields.py "$t"
done | tee $O/sweep.txt
