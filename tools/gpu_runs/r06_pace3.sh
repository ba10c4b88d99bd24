#!/bin/bash
# round 6: the paced schedule, lower periods for K2f and a first look at K1 (headline: sinebuf, 65 536 voices)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace3; mkdir -p $O
for r in 1 2; do for p in 0 46 48 50 52 54 56 58 60 64; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --steps 512 --warmup 64 --kernel-events off --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "K1 osc_pace=$p r$r"
[ $p = 0 ] && continue
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=2 --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA diet r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeB r$r"
done; done | tee $O/ab.txt
