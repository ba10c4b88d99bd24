#!/bin/bash
# round 6: the default bench line (few launches per entry: the controllers have to be there at once), twice
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06short; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default2.json 2> $O/bench_default2.err
wc -c $O/*.json
