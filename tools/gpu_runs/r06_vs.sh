#!/bin/bash
# round 6: K2f's store-stream knobs swept again on the current kernel (config 3, 65 536 voices; the automatic rule dates from round 3)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06vs; mkdir -p $O
for st in 0 1 2 3 4 5; do for x in 1 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 256 --warmup 64 --kernel-events off --tune voice_store=$st --tune voice_xcd=$x 2>> $O/err.log | python tools/line_fields.py "voice_store=$st voice_xcd=$x"
done; done | tee $O/sweep.txt
for st in 0 3 4 5; do
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 256 --warmup 64 --kernel-events off --tune voice_mix_store=$st 2>> $O/err.log | python tools/line_fields.py "MIX voice_mix_store=$st"
done | tee -a $O/sweep.txt
