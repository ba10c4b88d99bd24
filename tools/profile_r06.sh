#!/bin/bash
# tools/profile_r06.sh [tag] -- rocprofv3 evidence for every bench.py workload (run via gpurun).  Round 6: + config3_mix (K2f with the
# fused mixdown), sample_bank (playAtSpeed over an HBM-resident sample); config2_tables is the pipelined step.
# Per workload three SEPARATE passes: --kernel-trace --stats; --pmc WRITE_SIZE; --pmc FETCH_SIZE (counters are never
# combined with a trace domain), plus an MFMA counter pass for the dense mel contraction and an un-profiled bench line.
# Summaries: tools/summarize_rocprof.py <tag> -> profiles/<tag>_<workload>_summary.md, profiles/pmc_traffic.json.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, trace-steps, pmc-steps, bench args...   (ONLY="config5 config4": just those workloads)
  name=$1; ts=$2; ps=$3; shift 3
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/kt -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps $ts --warmup 3 "$@" > $OUT/$name.kt.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/pmc_w -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps $ps --warmup 2 "$@" > $OUT/$name.w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/pmc_r -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps $ps --warmup 2 "$@" > $OUT/$name.r.log 2>&1
  python $R/bench.py --no-cpu-baseline --verbose "$@" > $OUT/$name.bench.json 2> $OUT/$name.bench.err
}
run config2 500 20 --no-extras
run config2_131072 300 20 --no-extras --voices 131072
run config2_mix 500 20 --mixdown fused
run config2_tables 100 10 --workload tables
run config3 512 20 --workload config3
run config3_mix 512 20 --workload config3 --mixdown fused
run config3_modB 128 8 --workload config3 --voice-mode 1   # (128 blocks = one whole gate cycle: the step depends on the envelope's stage)
run sample_bank 100 10 --workload sample_bank
run config4 6 3 --workload config4
run config4_walk 6 3 --workload config4 --mfcc-method walk
run config4_mfma 6 3 --workload config4 --mfcc-method mfma
run config4_gemm 6 3 --workload config4 --mfcc-method mfma-gemm --mfma-fullk
run config5 6 3 --workload config5
# the matrix pipe of the fused kernel (default form = the matrix form) and the chip clock while it runs
if [ -z "$ONLY" ]; then
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/config4_mfma/pmc_mfma -o b -- \
    python $R/bench.py --no-cpu-baseline --kernel-events off --steps 3 --warmup 2 --workload config4 --mfcc-method mfma > $OUT/config4_mfma.mfma.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/config4_mfma/pmc_clk -o b -- \
    python $R/bench.py --no-cpu-baseline --kernel-events off --steps 3 --warmup 2 --workload config4 --mfcc-method mfma > $OUT/config4_mfma.clk.log 2>&1
fi
# fp64 flops of the per-sample-modulated voice, over one whole gate cycle
if [ -z "$ONLY" ] || echo " $ONLY " | grep -q " config3_modB "; then
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --output-format csv -d $OUT/config3_modB/pmc_f64 -o b -- \
    python $R/bench.py --no-cpu-baseline --kernel-events off --steps 128 --warmup 0 --workload config3 --voice-mode 1 > $OUT/config3_modB.f64.log 2>&1
fi
# keep what travels back small: the counter CSVs of torch's start-up kernels are not needed
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*.db" -delete; du -sh $OUT
cd $R
ls $OUT
