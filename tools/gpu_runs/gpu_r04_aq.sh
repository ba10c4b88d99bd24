#!/bin/bash
# round 4: pulse by the sign of phase - duty, triangle's operand select -- parity on the corners, then every waveform
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04aq
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_osc.py tests/test_gpu_fullparity.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for r in 1 2; do MODE=one REPS=300 timeout 300 python tools/sweep_heavy_osc.py 2 4 5 6 7 11 2>&1 | grep "^wf" | tr '\n' ' '; echo; done
