#!/bin/bash
# round 6: K2f's steady paths as runs (one tight loop over the chunks that keep the gate's class): parity, then timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06runs; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_voice.py tests/test_gpu_fullparity.py -q -x -k "voice or config3" 2>&1 | tail -5 > $O/t.log
tail -3 $O/t.log
for r in 1 2; do
for d in 1 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=$d 2>> $O/err.log | python tools/line_fields.py "modeA diet=$d r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=$d --tune voice_store=5 2>> $O/err.log | python tools/line_fields.py "modeA pair rows nt diet=$d r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=$d --tune voice_store=4 2>> $O/err.log | python tools/line_fields.py "modeA pair rows sc1 diet=$d r$r"
timeout 300 python bench.py --workload config3 --voices 32768 --no-cpu-baseline --no-extras --no-configs --steps 256 --warmup 64 --kernel-events off --tune voice_diet=$d 2>> $O/err.log | python tools/line_fields.py "modeA 32768 diet=$d r$r"
done
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeB gate cycle r$r"
done | tee $O/ab.txt
