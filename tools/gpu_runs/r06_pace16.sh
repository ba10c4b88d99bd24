#!/bin/bash
# round 6: the sample players on a fixed-period schedule (knob smp_pace), tools/bench_banks.py's three sample lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace16; mkdir -p $O
for p in 0 48 52 56 60 64 100 108 116 124; do
timeout 300 python tools/bench_banks.py smp_pace=$p 2>/dev/null | grep "^sample" | sed "s/^/smp_pace=$p /"
done | tee $O/ab.txt
