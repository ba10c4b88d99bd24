# tools/ab_many.sh LIB... -- config 4 kernel_ms (exact, tolerance) for several builds on the same box, two rounds
cd $GRAFT_REPO_ROOT
for round in 1 2; do for lib in "$@"; do for ex in 1 0; do
  MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --workload config4 --no-cpu-baseline --no-extras --steps 20 --warmup 5 --tune fft_exact=$ex 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib exact=$ex round $round kernel_ms', d['roofline'].get('kernel_ms'))
"
done; done; done
