#!/bin/bash
# round 4: SQ counters of the VALU-heavy oscillators after the diet (three passes, <= 4 counters each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wf in 0 9 10; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    MODE=one REPS=20 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/sq_wf$wf/g$i -o k -- python $R/tools/sweep_heavy_osc.py $wf > $O/sq_wf$wf.g$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for wf in (0, 9, 10):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/r04z/sq_wf%d/g*/*counter_collection.csv" % wf):
        for r in csv.DictReader(open(f)):
            if "osc_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("## wf", wf)
    for k in sorted(acc):
        v = sum(acc[k]) / len(acc[k])
        print("%-28s %.4g per launch, %.2f per wavefront-sample" % (k, v, v / (65536 * 512 / 64)))
PY
