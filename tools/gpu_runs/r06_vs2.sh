#!/bin/bash
# round 6: K2f at 65 536 voices, the automatic store stream (pair rows, non-temporal) against non-temporal 8-byte stores, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06vs2; mkdir -p $O
for r in 1 2 3 4; do for st in 0 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_store=$st 2>> $O/err.log | python tools/line_fields.py "voice_store=$st r$r"
done; done | tee $O/ab.txt
