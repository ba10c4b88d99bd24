#!/bin/bash
# round 6: what the driver runs at the end of the round -- the whole GPU suite, smoke(), the default bench line (+ a second one), N = 2 on one GPU
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06full; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default2.json 2> $O/bench_default2.err
timeout 600 python bench.py --gpus 2 --share-gpu --steps 64 --warmup 8 --no-cpu-baseline 2> $O/bench_n2.err | grep "^{" > $O/bench_n2.json
tail -n 6 $O/pytest_gpu.log; cat $O/smoke.log | tail -2; wc -c $O/bench_default.json $O/bench_n2.json
