#!/bin/bash
# tools/profile_configs.sh <tag> [cfg ...] -- rocprofv3 kernel-trace stats + tool output for configs 3, 4, 5 (run via gpurun).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_configs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
shift
CFGS=${@:-voice spectral grains banks mix}
for cfg in $CFGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg -o k -- python $R/tools/bench_$cfg.py > $OUT/$cfg.profiled.log 2>&1
  python $R/tools/bench_$cfg.py > $OUT/$cfg.log 2>&1
done
cd $R
