#!/bin/bash
# round 2, call l: fused FFT+MFCC with hand-packed butterflies / post-pass and the pipelined mel walk: parity, A/B of the two
# forms, and SQ counters of the 8-wave form (VALU busy model)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullparity.py tests/test_gpu_convolve.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for a in "--tune fused_waves16=0" "--tune fused_waves16=1" "--mfcc-method mfma --mfma-fullk"; do
echo "== bench.py --workload config4 $a" >> $O/bench.log
timeout 600 python bench.py --no-cpu-baseline --workload config4 $a >> $O/bench.log 2>> $O/bench.err
done
grep -o '"ms_per_step": [0-9.]*\|"kernels": {[^}]*}[^}]*}' $O/bench.log
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps 3 --warmup 2 --workload config4 --tune fused_waves16=0 > $O/pmc_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r02l/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'fft_mfcc' in r['Kernel_Name']:
            k = r['Counter_Name']; acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for k, (v, n) in acc.items():
        print("%-28s per launch %.4g (%d launches)" % (k, v / max(n, 1), n))
PY
