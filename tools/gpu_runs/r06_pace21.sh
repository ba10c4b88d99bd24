#!/bin/bash
# round 6: K2f mode A paced, natural against the automatic (XCD-contiguous from 98 304 voices) workgroup numbering, by bank size
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace21; mkdir -p $O; rm -f $O/err.log
for V in 98304 131072 196608 262144 393216 524288; do
base=$(( V * 56 / 65536 ))
for x in 0 1; do for f in 0 96 102 108; do
if [ $f = 0 ]; then p=1; else p=$(( base * f / 100 )); fi
timeout 300 python bench.py --workload config3 --voices $V --no-cpu-baseline --no-extras --no-configs --steps 160 --warmup 40 --kernel-events off --tune voice_pace=$p --tune voice_xcd=$x 2>> $O/err.log | python tools/line_fields.py "V=$V voice_xcd=$x pace=$p"
done; done; done | tee $O/ab.txt
