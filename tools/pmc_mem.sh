#!/bin/bash
# tools/pmc_mem.sh <tag> <tool> -- memory-side counters for one tools/bench_<tool>.py (run via gpurun): separate rocprofv3 --pmc passes.
TAG=${1:-mem}; TOOL=${2:-speedplayer}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  i=$((i+1))
  REPS=2 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o k -- python $R/tools/bench_$TOOL.py > $OUT/g$i.log 2>&1
done
cd $R
