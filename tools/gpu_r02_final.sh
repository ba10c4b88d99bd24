#!/bin/bash
# round 2: the evidence run -- full GPU suite, rocprofv3 passes of every bench workload (tools/profile_r02.sh), the per-kernel
# tool benches, the bank-size sweep and the SQ counters of the fused FFT+MFCC kernel.  Outputs under gpurun_out/final/ and
# gpurun_out/prof_r02/; tools/summarize_r02.py + the copy lines in the commit bring them to profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash tools/profile_r02.sh r02 > $O/profile.log 2>&1
timeout 600 python tools/bench_banks.py > $O/banks.txt 2> $O/banks.err
timeout 600 python tools/bench_waveforms.py > $O/waveforms.txt 2> $O/waveforms.err
timeout 900 python tools/sweep_banks.py > $O/sweep.md 2> $O/sweep.err
timeout 600 python tools/sweep_time_parts.py > $O/time_parts.txt 2> $O/time_parts.err
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o b -- \
      python $R/bench.py --no-cpu-baseline --kernel-events off --steps 3 --warmup 2 --workload config4 > $O/pmc_$tag.log 2>&1
done
cd $R
python - > $O/config4_sq.txt <<'PY'
import csv, glob, collections
print("SQ counters of fft_mfcc_kernel, one launch = 1 048 576 frames (rocprofv3 --pmc, three separate passes)")
for f in sorted(glob.glob('gpurun_out/final/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'fft_mfcc' in r['Kernel_Name']:
            k = r['Counter_Name']; acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for k, (v, n) in acc.items():
        print("%-28s %.4g per launch" % (k, v / max(n, 1)))
PY
cat $O/config4_sq.txt; cat $O/banks.txt; cat $O/waveforms.txt | tail -14; tail -12 $O/sweep.md
