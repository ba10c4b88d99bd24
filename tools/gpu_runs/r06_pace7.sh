#!/bin/bash
# round 6: K2f as it now ships (short instruction stream everywhere, the controlled schedule at 65 536 voices): tests, then the forms and sizes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06pace7; mkdir -p $O; rm -f $O/err.log
timeout 2400 python -m pytest tests/test_gpu_voice.py tests/test_gpu_fullparity.py tests/test_gpu_host.py tests/test_gpu_dropin.py -q -x -k "voice or config3 or polysynth or monosynth or synth" 2>&1 | tail -4 > $O/t.log; tail -3 $O/t.log
for r in 1 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeA r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_pace=1 2>> $O/err.log | python tools/line_fields.py "modeA unpaced r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_pace=1 --tune voice_diet=1 2>> $O/err.log | python tools/line_fields.py "modeA unpaced, round-5 stream r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeB r$r"
for V in 49152 81920 98304 131072; do for p in 1 0; do
timeout 300 python bench.py --workload config3 --voices $V --no-cpu-baseline --no-extras --no-configs --steps 256 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "modeA V=$V pace=$p r$r"
done; done
done | tee $O/ab.txt
grep "^pace" $O/err.log | tail -12
