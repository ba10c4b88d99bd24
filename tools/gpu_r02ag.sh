#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o b -- python $R/tools/sweep_grains_general.py 128 > $O/pmc_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r02ag/pmc_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'granular_line_kernel' in r['Kernel_Name']:
            k = r['Counter_Name']; acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for k, (v, n) in acc.items():
        print("%-28s %.4g per launch (%d launches)" % (k, v / max(n, 1), n))
PY
