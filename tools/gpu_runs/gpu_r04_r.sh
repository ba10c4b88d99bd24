#!/bin/bash
# round 4: tables kernel with non-temporal DMA + the shorter marks step (A/B on one box); libmaxicalib.so through bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osctab.py tests/test_bench_launch.py -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for round in 1 2; do for lib in libmaxigpu.so ab_tab_aux0.so; do
  echo "== $lib round $round" | tee -a $O/osctab.txt
  MXG_LIB=$R/maximilian_amd/$lib timeout 300 python tools/bench_osctab.py 131072,262144 2>&1 | grep "mixdown only" | tee -a $O/osctab.txt
done; done
timeout 300 python tools/write_ceiling.py > $O/write_ceiling.log 2>&1; tail -3 $O/write_ceiling.log
