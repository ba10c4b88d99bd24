#!/bin/bash
# tools/profile_r04.sh [tag] -- rocprofv3 evidence for every bench.py workload (run via gpurun).
# Per workload three SEPARATE passes: --kernel-trace --stats; --pmc WRITE_SIZE; --pmc FETCH_SIZE (counters are never
# combined with a trace domain), plus an MFMA counter pass for the dense mel contraction and an un-profiled bench line.
# Summaries: tools/summarize_rocprof.py <tag> -> profiles/<tag>_<workload>_summary.md, profiles/pmc_traffic.json.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, trace-steps, pmc-steps, bench args...
  name=$1; ts=$2; ps=$3; shift 3
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/kt -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps $ts --warmup 3 "$@" > $OUT/$name.kt.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/pmc_w -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps $ps --warmup 2 "$@" > $OUT/$name.w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/pmc_r -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps $ps --warmup 2 "$@" > $OUT/$name.r.log 2>&1
  python $R/bench.py --no-cpu-baseline "$@" > $OUT/$name.bench.json 2> $OUT/$name.bench.err
}
run config2 500 20 --no-extras
run config2_131072 300 20 --no-extras --voices 131072
run config2_196608 200 20 --no-extras --voices 196608
run config2_mix 500 20 --mixdown fused
run config2_tables 100 10 --workload tables
run config3 512 20 --workload config3
run config4 6 3 --workload config4
run config4_mfma 6 3 --workload config4 --mfcc-method mfma --mfma-fullk
run config5 6 3 --workload config5
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/config4_mfma/pmc_mfma -o b -- \
    python $R/bench.py --no-cpu-baseline --kernel-events off --steps 3 --warmup 2 --workload config4 --mfcc-method mfma --mfma-fullk > $OUT/config4_mfma.mfma.log 2>&1
cd $R
ls $OUT
