#!/bin/bash
# round 6, call z: the N > 1 step on one GPU (K1m + grouped mix queue) under the batch depth M (blocks per reduce)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06z; mkdir -p $O
for r in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 1024 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K1 r$r"
for m in 8 16 32 64; do
timeout 300 python bench.py --mixdown fused --mix-depth $m --no-cpu-baseline --steps 1024 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K1m M=$m r$r"
done; done | tee $O/ab.txt
