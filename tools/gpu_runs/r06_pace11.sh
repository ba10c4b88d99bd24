#!/bin/bash
# round 6: K2f on the controlled schedule at every bank size (voice_pace 0 = automatic against 1 = never), the three forms
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06pace11; mkdir -p $O; rm -f $O/err.log
for V in 16384 32768 40960 49152 65536 81920 98304 131072 196608 262144 524288; do for p in 1 0; do
timeout 300 python bench.py --workload config3 --voices $V --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "modeA V=$V pace=$p"
done; done | tee $O/ab.txt
for V in 131072; do for p in 1 0; do
timeout 300 python bench.py --workload config3 --voices $V --mixdown fused --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "modeA+mix V=$V pace=$p"
timeout 300 python bench.py --workload config3 --voices $V --voice-mode 1 --no-cpu-baseline --no-extras --no-configs --steps 128 --warmup 128 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "modeB V=$V pace=$p"
done; done | tee -a $O/ab.txt
grep "^pace" $O/err.log | tail -16
