#!/bin/bash
# round 6: K1 on the controlled schedule by bank size and waveform (osc_pace 0 = automatic against 1 = the round-4 launch rules)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06pace13; mkdir -p $O; rm -f $O/err.log

for V in 69632 73728 81920 98304 131072 196608 262144 393216 524288; do for p in 1 0; do
timeout 300 python bench.py --voices $V --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "sinebuf V=$V osc_pace=$p"
done; done | tee $O/ab.txt
for wf in saw square triangle pulse phasor; do for p in 1 0; do
timeout 300 python bench.py --voices 131072 --waveform $wf --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off --tune osc_pace=$p 2>> $O/err.log | python tools/line_fields.py "$wf V=131072 osc_pace=$p"
done; done | tee -a $O/ab.txt
grep "^pace" $O/err.log | tail -16
for r in 1 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeA r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeB r$r"
timeout 300 python bench.py --workload config3 --voices 131072 --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeA 131072 r$r"
done | tee -a $O/ab.txt
