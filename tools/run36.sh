cd $GRAFT_REPO_ROOT
for V in 1024 4096 16384 32768; do for sp in 1 2 4 8; do
  timeout 200 python bench.py --voices $V --no-cpu-baseline --no-extras --steps 300 --warmup 30 --tune osc_split=$sp 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('V=$V split=$sp ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))
"
done; done
