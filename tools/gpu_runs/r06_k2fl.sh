#!/bin/bash
# round 6: K2f plain mode A at 65 536 voices with the instruction diet: every store flavour and XCD numbering, against the build without the diet
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06k2fl; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_oldk2f.so; do for st in 1 2 3 4 5; do for x in 1 2; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-extras --no-configs --steps 256 --warmup 64 --kernel-events off --tune voice_store=$st --tune voice_xcd=$x 2>> $O/err.log | python tools/line_fields.py "$lib store=$st xcd=$x r$r"
done; done; done; done | tee $O/ab.txt
