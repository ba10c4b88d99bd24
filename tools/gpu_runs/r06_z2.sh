#!/bin/bash
# round 6, call z2: the same at the run lengths the bench uses (400 steps: configs.config2_mixdown; 20 and 64: driver-style N > 1 lines, two ranks on one GPU)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06z2; mkdir -p $O
for r in 1 2; do for m in 16 32 64; do for st in 400 2000; do
timeout 300 python bench.py --mixdown fused --mix-depth $m --no-cpu-baseline --steps $st --warmup 50 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K1m M=$m steps=$st r$r"
done; done; done | tee $O/ab.txt
for m in 16 32; do
timeout 600 python bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-configs --mix-depth $m 2>> $O/err.log | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('N=2 shared, K=20, M=$m: ms_per_step', d['ms_per_step'], 'without_reduce', d.get('step_ms_without_reduce'), 'eff', d.get('per_gpu_efficiency'))"
done | tee -a $O/ab.txt
