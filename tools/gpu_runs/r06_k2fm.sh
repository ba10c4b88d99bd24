#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06k2fm; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_voice.py tests/test_gpu_fullparity.py tests/test_gpu_host.py tests/test_gpu_dropin.py -q -x -k "voice or config3 or polysynth or monosynth or synth" 2>&1 | tail -8 > $O/t.log
tail -6 $O/t.log
for r in 1 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeA r$r"
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeB gate cycle r$r"
timeout 300 python bench.py --workload config3 --voices 32768 --no-cpu-baseline --no-extras --no-configs --steps 256 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modeA 32768 r$r"
done | tee $O/ab.txt
