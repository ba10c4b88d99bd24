# tools/ab_osc.sh WAVEFORM LIB... -- config 2 with another waveform: kernel_ms per build on the same box, three rounds
cd $GRAFT_REPO_ROOT
wf=$1; shift
for round in 1 2 3; do for lib in "$@"; do
  MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --waveform $wf --no-cpu-baseline --no-extras --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib $wf round $round kernel_ms', d['roofline'].get('kernel_ms'))
"
done; done
