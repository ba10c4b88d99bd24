#!/bin/bash
# round 6: sinebuf between 65 536 and 122 880 voices: the round-4 launch rules against the paced simple launch (tolerant controller)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06mid; mkdir -p $O; rm -f $O/err.log
for V in 69632 73728 77824 81920 86016 90112 98304 106496 114688; do for f in 122880 65536; do
MXG_PACE_SINEBUF_FROM=$f timeout 600 python bench.py --voices $V --steps 480 --warmup 64 --no-configs --no-extras --no-cpu-baseline --kernel-events off 2>> $O/err.log | python tools/line_fields.py "sinebuf V=$V paced from $f"
done; done | tee $O/ab.txt
