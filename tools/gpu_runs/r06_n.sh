#!/bin/bash
# round 6, call n: mode B with the single sine polynomial (parity + time), the polysynth host with the mixdown on the device
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host.py tests/test_gpu_voice.py -x -q 2>&1 | tail -4 > $O/t_a.log
timeout 1500 python -m pytest tests/test_gpu_fullparity.py -x -q -k "config3_128" -s 2>&1 | grep -a -E "mode B|passed|failed" > $O/t_b.log
for r in 1 2 3; do
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 256 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modB r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 256 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "modA r$r"
done | tee $O/ab.txt
cat $O/t_a.log $O/t_b.log
