#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_sample.py tests/test_gpu_spectral.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python tools/bench_waveforms.py > $O/waveforms.txt 2>&1
python tools/sweep_banks.py > $O/sweep.md 2> $O/sweep.err
cat $O/waveforms.txt | head -16; tail -14 $O/sweep.md; tail -3 $O/sweep.err
