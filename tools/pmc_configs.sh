#!/bin/bash
# tools/pmc_configs.sh <tag> [cfg ...] -- HBM traffic counters for the per-config tools (run via gpurun).
# Two separate rocprofv3 passes per tool (--pmc WRITE_SIZE, then --pmc FETCH_SIZE), never combined with any
# trace domain other than the implicit kernel dispatch records.  Output: gpurun_out/pmc_<tag>/<cfg>/{w,r}/.
TAG=${1:-r01}
shift
CFGS=${@:-voice spectral grains banks mix}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in $CFGS; do
  for c in WRITE_SIZE FETCH_SIZE; do
    d=$OUT/$cfg/$( [ $c = WRITE_SIZE ] && echo w || echo r )
    REPS=2 rocprofv3 --pmc $c --output-format csv -d $d -o k -- python $R/tools/bench_$cfg.py > $OUT/$cfg.$c.log 2>&1
  done
done
cd $R
