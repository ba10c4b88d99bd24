#!/bin/bash
# round 6, call p: K1m's producer / consumer form taken apart again on the shipped build (consumer arithmetic, tile writes), bare kernel loops
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06p; mkdir -p $O
for r in 1 2; do for lib in maximilian_amd/libmaxigpu.so build/ab/ab_pcnc.so build/ab/ab_pcskel.so; do
MXG_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/bench_k1m_tail.py 2>> $O/err.log | sed "s|^|$lib r$r |"
done; done | tee $O/ab.txt
