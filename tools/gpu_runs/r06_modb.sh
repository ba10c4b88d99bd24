#!/bin/bash
# round 6: mode B's small-angle coefficients (lores_coeffs_sin_small): parity, then interleaved with a build without them
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06modb; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_voice.py tests/test_gpu_fullparity.py -q -x -k "voice or config3" 2>&1 | tail -5 > $O/t.log
tail -3 $O/t.log
for r in 1 2 3; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_nosmall.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeB gate cycle r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --voice-mode 1 --mixdown fused --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeB+mix r$r"
done; done | tee $O/ab.txt
