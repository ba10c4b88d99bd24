#!/bin/bash
# tools/pmc_sq.sh <tag> [cfg ...] -- shader-core counters for the per-config tools (run via gpurun).  Three separate
# rocprofv3 passes per tool, <= 4 SQ counters each, never combined with a trace domain.
TAG=${1:-r01}
shift
CFGS=${@:-voice spectral grains mix}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in $CFGS; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    REPS=2 rocprofv3 --pmc $grp --output-format csv -d $OUT/$cfg/g$i -o k -- python $R/tools/bench_$cfg.py > $OUT/$cfg.g$i.log 2>&1
  done
done
cd $R
