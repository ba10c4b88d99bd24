#!/bin/bash
# round 6, call a: the new K2f mixdown form (parity), the tightened matrix-pipe guards, the refused-call patch; first bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_voice.py -x -q -k "mix_fused" 2>&1 | tail -15 > $O/t_voice.log
timeout 900 python -m pytest tests/test_gpu_spectral.py -x -q 2>&1 | tail -15 > $O/t_spectral.log
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -k "refused or granular" 2>&1 | tail -15 > $O/t_dropin.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --gpus 2 --share-gpu --steps 64 --warmup 8 --no-cpu-baseline > $O/bench_n2.json 2> $O/bench_n2.err
tail -3 $O/t_voice.log $O/t_spectral.log $O/t_dropin.log; wc -c $O/bench_default.json $O/bench_n2.json; tail -5 $O/bench_default.err $O/bench_n2.err
