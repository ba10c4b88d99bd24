#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06vs4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_voice.py tests/test_gpu_fullparity.py -x -q -k "not config4 and not config5 and not config2" 2>&1 | tail -3 > $O/t.log
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/err.log
tail -2 $O/t.log; wc -c $O/bench_default.json
