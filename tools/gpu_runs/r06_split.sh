#!/bin/bash
# round 6: K2f on two wavefronts per 64 voices (voice_split = 2) against the one-wavefront kernel: parity, then interleaved timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06split; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_voice.py -q --tb=short -k "two_stage" 2>&1 | tail -60 > $O/t.log
tail -40 $O/t.log
for r in 1 2; do for sp in 1 2; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_split=$sp 2>> $O/err.log | python tools/line_fields.py "modeA voice_split=$sp r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_split=$sp --tune voice_store=4 2>> $O/err.log | python tools/line_fields.py "modeA pairnt voice_split=$sp r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_split=$sp 2>> $O/err.log | python tools/line_fields.py "modeB voice_split=$sp r$r"
done; done | tee $O/ab.txt
