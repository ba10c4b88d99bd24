#!/bin/bash
# round 6: the controller judging a launch by the MAJORITY of its reporters: K2f again, saw, and sinebuf (MXG_SINEBUF_PACED=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06pace15; mkdir -p $O; rm -f $O/err.log
for r in 1 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeA r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeB r$r"
timeout 300 python bench.py --workload config3 --voices 131072 --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeA 131072 r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 32 --warmup 8 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f short run (8 + 32) r$r"
for V in 98304 131072 196608 262144; do
timeout 300 python bench.py --voices $V --waveform saw --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "saw V=$V r$r"
MXG_SINEBUF_PACED=1 timeout 300 python bench.py --voices $V --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "sinebuf paced V=$V r$r"
timeout 300 python bench.py --voices $V --no-cpu-baseline --no-extras --no-configs --steps 320 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "sinebuf plan V=$V r$r"
done; done | tee $O/ab.txt
grep "^pace" $O/err.log | tail -20
