#!/bin/bash
# round 6: K2f with a pause per steady chunk (knob voice_pace = number of s_sleep 1), every form, 65 536 voices
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace; mkdir -p $O
for r in 1 2; do for p in 0 8 10 11 12 13 14 16 20; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=2 --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=2 --tune voice_store=5 --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA pair rows nt r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeB r$r"
done; done | tee $O/ab.txt
