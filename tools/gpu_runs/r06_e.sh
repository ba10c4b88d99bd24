#!/bin/bash
# round 6, call e: config 4 with two frame sets in flight (PF 2, the build) against one (ab_pf1), same box; parity of the new build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
bash tools/ab_many.sh maximilian_amd/libmaxigpu.so build/ab/ab_pf1.so 2>&1 | tee $O/pf.txt
timeout 900 python -m pytest tests/test_gpu_spectral.py -x -q 2>&1 | tail -5 > $O/t_spectral.log
timeout 1500 python -m pytest tests/test_gpu_fullparity.py -x -q -k "config4" -s 2>&1 | tail -12 > $O/t_full.log
tail -n 5 $O/t_spectral.log $O/t_full.log
