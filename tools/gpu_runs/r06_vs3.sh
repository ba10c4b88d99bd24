#!/bin/bash
# round 6: the same on another box, + mode B and the mixdown form
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06vs3; mkdir -p $O
for r in 1 2 3; do for st in 0 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_store=$st 2>> $O/err.log | python tools/line_fields.py "A voice_store=$st r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_store=$st 2>> $O/err.log | python tools/line_fields.py "B voice_store=$st r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_mix_store=$st 2>> $O/err.log | python tools/line_fields.py "MIX voice_mix_store=$st r$r"
done; done | tee $O/ab.txt
