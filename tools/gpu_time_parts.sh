#!/bin/bash
# GPU run: parity tests of the time-part kernels + per-bank times + the osc_split / smp_split sweeps (results in gpurun_out/exp)
O=gpurun_out/exp; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_envgen.py tests/test_gpu_sample.py tests/test_gpu_osc.py tests/test_gpu_sampler.py tests/test_gpu_edges.py -m gpu -q -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/bench_banks.py > $O/banks.txt 2>&1
grep -i "envgen\|sample\|env adsr" $O/banks.txt
timeout 300 python tools/bench_waveforms.py > $O/waveforms.txt 2>&1
grep -i "sinewave\|coswave\|sinebuf4\|sawn" $O/waveforms.txt
python tools/sweep_time_parts.py > $O/knobs.txt 2>&1
cat $O/knobs.txt
