#!/bin/bash
# re-run only the un-profiled bench lines of tools/profile_r02.sh (same arguments), into gpurun_out/prof_r02/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd $R
run() { name=$1; shift; python bench.py "$@" > $OUT/$name.bench.json 2> $OUT/$name.bench.err; tail -c 400 $OUT/$name.bench.json | head -c 200; echo; }
run config2
run config2_mix --mixdown fused
run config3 --workload config3
run config4 --workload config4
run config4_mfma --workload config4 --mfcc-method mfma --mfma-fullk
run config5 --workload config5
