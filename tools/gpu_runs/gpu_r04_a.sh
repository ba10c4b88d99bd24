#!/bin/bash
# tools/gpu_r04_a.sh -- round 4, first GPU call: parity of the new kernels, then the measurements that decide their defaults.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_comm.py -x -q -m gpu > $O/pytest_osc_comm.log 2>&1
echo "pytest rc=$?" >> $O/pytest_osc_comm.log
tail -5 $O/pytest_osc_comm.log
for w in 0 128 256; do
  timeout 300 python bench.py --mixdown fused --no-cpu-baseline --tune osc_mix_win=$w > $O/bench_mix_win$w.json 2> $O/bench_mix_win$w.err
done
timeout 300 python bench.py --mixdown fused --no-cpu-baseline --mix-only > $O/bench_mix_only.json 2> $O/bench_mix_only.err
timeout 300 python bench.py --mixdown fused --no-cpu-baseline --voices 131072 > $O/bench_mix_131072.json 2> $O/bench_mix_131072.err
timeout 600 python bench.py --gpus 2 --share-gpu --no-cpu-baseline --steps 400 --warmup 50 > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python tools/sweep_osc_persist.py --out $O/osc_persist.md > $O/sweep.log 2>&1
timeout 600 python -m pytest tests/test_bench_launch.py -x -q -m gpu > $O/pytest_bench.log 2>&1
echo "pytest rc=$?" >> $O/pytest_bench.log
grep -h '"ms_per_step"' $O/bench_mix_win*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['kernels'])
"
tail -3 $O/pytest_bench.log
