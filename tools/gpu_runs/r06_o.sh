#!/bin/bash
# round 6, call o: play() with heads between elements on the row-load path: tests + tools/bench_banks.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sample.py tests/test_gpu_sampler.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -4 > $O/t.log
timeout 1200 python tools/bench_banks.py 2>> $O/err.log | grep -i "sample" | tee $O/banks.txt
tail -n 3 $O/t.log
