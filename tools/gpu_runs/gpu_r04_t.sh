#!/bin/bash
# round 4: pair-row streams of env / delay / sample / envgen -- parity, then the per-bank times per rw_store setting
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04t
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_rw_store.py tests/test_gpu_voice.py tests/test_gpu_sample.py tests/test_gpu_envgen.py tests/test_gpu_filter2.py tests/test_gpu_edges.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rw in 1 2 3 4 0; do
  timeout 300 python tools/bench_banks.py --rw $rw > $O/banks_rw$rw.txt 2>&1
done
paste -d'|' <(cut -c1-38 $O/banks_rw1.txt) <(cut -c27-38 $O/banks_rw2.txt) <(cut -c27-38 $O/banks_rw3.txt) <(cut -c27-38 $O/banks_rw4.txt) <(cut -c27-38 $O/banks_rw0.txt)
timeout 300 python tools/bench_waveforms.py > $O/waveforms.txt 2>&1; cat $O/waveforms.txt
