#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_voice.py tests/test_gpu_grains.py -x -q -k "mix_fused or rendered_again" 2>&1 | tail -8 | tee $O/t.log
