#!/bin/bash
# tools/profile_r03.sh [tag] -- rocprofv3 evidence for every bench.py workload (run via gpurun).
# Per workload three SEPARATE passes: --kernel-trace --stats; --pmc WRITE_SIZE; --pmc FETCH_SIZE (counters are never
# combined with a trace domain), plus an MFMA counter pass for the dense mel contraction and an un-profiled bench line.
TAG=${1:-r03}
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
run config2_mix 500 20 --mixdown fused
run config3 512 20 --workload config3
run config4 6 3 --workload config4
run config4_tol 6 3 --workload config4 --tune fft_exact=0
run config4_mfma 6 3 --workload config4 --mfcc-method mfma --mfma-fullk
run config5 6 3 --workload config5
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/config4_mfma/pmc_mfma -o b -- \
    python $R/bench.py --no-cpu-baseline --kernel-events off --steps 3 --warmup 2 --workload config4 --mfcc-method mfma --mfma-fullk > $OUT/config4_mfma.mfma.log 2>&1
# SQ counters of the fused FFT+MFCC kernel, exact and tolerance mode (three separate --pmc passes each, no trace domain)
for mode in 1 0; do
  i=0
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/sq_exact${mode}/p$i -o b -- \
        python $R/bench.py --no-cpu-baseline --kernel-events off --steps 3 --warmup 2 --workload config4 --tune fft_exact=$mode > $OUT/sq_exact${mode}_p$i.log 2>&1
  done
done
cd $R
python - > $OUT/config4_sq.md <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT_DIR", "")
print("# SQ counters of `fft_mfcc_kernel`, one launch = 1 048 576 frames (rocprofv3 --pmc, three separate passes per mode; tools/profile_r03.sh)\n")
print("Round 2's kernel (profiles/r02_config4_sq.md, 1.43-1.45 ms): INSTS_VALU 4.647e8, ACTIVE_INST_VALU 4.72e8, WAVE_CYCLES 1.496e9, "
      "INSTS_LDS 7.891e7, LDS_IDX_ACTIVE 4.843e8, LDS_BANK_CONFLICT 1.23e8, WAIT_INST_ANY 3.17e8, INSTS_SALU 9.086e7.\n")
for mode, name in (("1", "exact (default)"), ("0", "tolerance mode (fft_exact = 0)")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob('gpurun_out/prof_*/sq_exact%s/**/*counter_collection.csv' % mode, recursive=True)):
        for r in csv.DictReader(open(f)):
            if 'fft_mfcc' in r['Kernel_Name']:
                k = r['Counter_Name']; acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    print("## %s\n\n| counter | per launch | per frame |\n|---|---|---|" % name)
    for k in sorted(acc):
        v = acc[k][0] / max(acc[k][1], 1)
        print("| %s | %.4g | %.1f |" % (k, v, v / 1048576.0))
    print()
PY
ls $OUT
