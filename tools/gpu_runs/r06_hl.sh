#!/bin/bash
# round 6: the headline (sinebuf, 65 536 voices) free-running against paced (MXG_PACE_SINEBUF_FROM=65536), inside the default line and alone
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06hl; mkdir -p $O; rm -f $O/err.log
for r in 1 2 3; do for f in 122880 65536; do
MXG_PACE_SINEBUF_FROM=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>> $O/err.log | python tools/line_fields.py "default line, sinebuf paced from $f r$r"
MXG_PACE_SINEBUF_FROM=$f timeout 600 python bench.py --steps 512 --warmup 64 --no-configs --no-extras --no-cpu-baseline --kernel-events off 2>> $O/err.log | python tools/line_fields.py "512 steps, sinebuf paced from $f r$r"
done; done | tee $O/ab.txt
grep "^pace\[wave" $O/err.log | tail -8
