#!/bin/bash
# round 6: K2f mode A at 262 144 voices on fixed periods, natural and XCD-contiguous numbering
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace20; mkdir -p $O; rm -f $O/err.log
for x in 0 1 2; do for p in 1 212 224 236 248 260; do
timeout 300 python bench.py --workload config3 --voices 262144 --no-cpu-baseline --no-extras --no-configs --steps 200 --warmup 40 --kernel-events off --tune voice_pace=$p --tune voice_xcd=$x 2>> $O/err.log | python tools/line_fields.py "V=262144 voice_xcd=$x pace=$p"
done; done | tee $O/ab.txt
