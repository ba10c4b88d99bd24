#!/bin/bash
# round 6, call d: config 4 ablations on the shipped kernel (what the two untried levers could save at most); the HBM-resident sample bank
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
bash tools/ab_many.sh maximilian_amd/libmaxigpu.so build/ab/ab_abl32.so build/ab/ab_abl4.so build/ab/ab_abl16.so build/ab/ab_abl2.so 2>&1 | tee $O/ablate.txt
timeout 600 python bench.py --workload sample_bank --steps 100 --warmup 20 > $O/sample_bank.json 2> $O/sample_bank.err
python tools/line_fields.py sample_bank < $O/sample_bank.json; tail -3 $O/sample_bank.err
