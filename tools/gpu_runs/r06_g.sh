#!/bin/bash
# round 6, call g: the modulated voice with one-sine coefficients (parity at full size), the voice tests, a default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_voice.py -x -q 2>&1 | tail -8 > $O/t_voice.log
timeout 1800 python -m pytest tests/test_gpu_fullparity.py -x -q -k "config3" -s 2>&1 | tail -12 > $O/t_full.log
timeout 900 python -m pytest tests/test_gpu_sample.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -5 > $O/t_misc.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
tail -n 6 $O/t_voice.log $O/t_full.log $O/t_misc.log; wc -c $O/bench_default.json
