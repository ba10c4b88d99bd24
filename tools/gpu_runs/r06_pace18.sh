#!/bin/bash
# round 6: maxiEnvGen banks on the paced schedule: fixed periods, then the controller (MXG_EG_PACED=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace18; mkdir -p $O
for p in 1 48 52 56 60 64; do
timeout 300 python tools/bench_banks.py eg_pace=$p 2>/dev/null | grep "^maxiEnvGen" | sed "s/^/eg_pace=$p /"
done | tee $O/ab.txt
for r in 1 2; do
MXG_EG_PACED=1 timeout 300 python tools/bench_banks.py 2>/dev/null | grep "^maxiEnvGen" | sed "s/^/controller /"
done | tee -a $O/ab.txt
