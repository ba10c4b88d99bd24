#!/bin/bash
# round 6: shader-core counters of K2f, one wavefront per 64 voices against the two-stage form (voice_split 1 / 2), separate --pmc passes
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_r06split; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sp in 1 2; do i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/s${sp}g$i -o k -- python $R/bench.py --workload config3 --no-cpu-baseline --no-extras --kernel-events off --steps 8 --warmup 4 --tune voice_split=$sp > $O/s${sp}g$i.log 2>&1
done; done
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/sq_r06split'
for sp in (1,2):
    acc=collections.defaultdict(list)
    for f in glob.glob(O+'/s%dg*/**/*counter_collection.csv'%sp, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'voice' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('voice_split=%d'%sp, {k: round(sum(v)/len(v)/ (1024*512),2) for k,v in sorted(acc.items())}, '(per wavefront-of-64-voices and sample; launches %d)'% (len(next(iter(acc.values()))) if acc else 0))
PY
