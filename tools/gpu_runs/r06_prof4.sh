#!/bin/bash
# round 6, late: rocprofv3 passes of the workloads the paced schedule changed + the per-kernel tool benches
cd $GRAFT_REPO_ROOT
ONLY="config2_131072 config3 config3_mix config3_modB" bash tools/profile_r06.sh r06 > gpurun_out/prof4.log 2>&1
tail -2 gpurun_out/prof4.log
export TMPDIR=/tmp
python tools/bench_banks.py > gpurun_out/r06_banks_late.txt 2>/dev/null
python tools/bench_waveforms.py > gpurun_out/r06_waveforms_late.txt 2>/dev/null

tail -3 gpurun_out/r06_banks_late.txt
