#!/bin/bash
# round 6: K2f on the controlled schedule with the natural numbering, by bank size (automatic against never), mode A and B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace22; mkdir -p $O; rm -f $O/err.log
timeout 900 python -m pytest tests/test_gpu_voice.py -q -x 2>&1 | tail -1
for V in 65536 98304 131072 196608 262144; do for p in 1 0; do
timeout 300 python bench.py --workload config3 --voices $V --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "modeA V=$V pace=$p"
done; done | tee $O/ab.txt
for V in 98304 131072; do for p in 1 0; do
timeout 300 python bench.py --workload config3 --voice-mode 1 --voices $V --no-cpu-baseline --no-extras --no-configs --steps 128 --warmup 128 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "modeB V=$V pace=$p"
done; done | tee -a $O/ab.txt
