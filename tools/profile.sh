#!/bin/bash
# tools/profile.sh <tag> -- rocprofv3 evidence for bench.py on the GPU box (run via gpurun).
# Three separate passes (kernel-trace+stats; PMC WRITE_SIZE; PMC FETCH_SIZE): counters are never
# combined with trace domains other than --kernel-trace.  Output under gpurun_out/prof_<tag>/.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- \
    python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o bench -- \
    python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_r -o bench -- \
    python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_r.log 2>&1
# un-profiled reference line for the same command
python $R/bench.py --steps 500 --warmup 50 > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
cd $R
