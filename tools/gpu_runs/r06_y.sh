#!/bin/bash
# round 6, call y: the shared (L2-resident) sample with the lean ring parts forced on (smp_ring 2), time parts 2 ... 6
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06y; mkdir -p $O
for r in 1 2; do
REPS=20 timeout 300 python tools/bench_speedplayer.py 2>> $O/err.log | sed "s|^|automatic r$r |"
for sp in 2 3 4 6; do
SMP_RING=2 SMP_SPLIT=$sp REPS=20 timeout 300 python tools/bench_speedplayer.py 2>> $O/err.log | sed "s|^|ring split=$sp r$r |"
done; done | tee $O/ab.txt
