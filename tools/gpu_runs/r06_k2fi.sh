#!/bin/bash
# round 6: K2f's instruction diet (one-add saw wrap, speculative release chunks, the steady-state test carried): parity, then interleaved with a build without
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06k2fi; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_voice.py -q -x -k "voice" 2>&1 | tail -5 > $O/t.log
tail -3 $O/t.log
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_oldk2f.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeA r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeA+mix r$r"
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "$lib modeB gate cycle r$r"
done; done | tee $O/ab.txt
