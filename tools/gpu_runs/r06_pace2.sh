#!/bin/bash
# round 6: K2f on a paced schedule (a chunk every P ticks of 10 ns; knob voice_pace: 1 off, >= 2 fixed P, 0 the controller), 65 536 voices
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_voice.py -q -x -k "voice" 2>&1 | tail -3 > $O/t.log; tail -2 $O/t.log
for r in 1 2; do for p in 1 58 60 62 64 66 68 70 0; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=2 --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA diet r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=1 --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA nodiet r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeB r$r"
done; done | tee $O/ab.txt
