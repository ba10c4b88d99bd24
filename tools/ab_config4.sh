# tools/ab_config4.sh LIB_A LIB_B -- config 4 (exact and tolerance mode) with two builds of libmaxigpu.so alternated on the SAME box
# (boxes differ by +-5 %: only numbers of one run compare).  Prints kernel_ms per build, three rounds.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
A=$1; B=$2
for round in 1 2 3; do for lib in $A $B; do for ex in 1 0; do
  MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config4 --no-cpu-baseline --no-extras --steps 20 --warmup 5 --tune fft_exact=$ex 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib exact=$ex round $round kernel_ms', d['roofline'].get('kernel_ms'))
"
done; done; done
