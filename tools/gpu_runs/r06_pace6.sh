#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06pace6; mkdir -p $O; rm -f $O/err.log
for r in 1 2; do for p in 0; do for d in 2 1; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=$d --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA diet=$d r$r"
done
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeB r$r"
done
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 32 --warmup 8 --kernel-events off --tune voice_diet=2 2>> $O/err.log | python tools/line_fields.py "short run (8 + 32) modeA diet r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 128 --warmup 16 --kernel-events off --tune voice_diet=2 2>> $O/err.log | python tools/line_fields.py "short run (16 + 128) modeA diet r$r"
done | tee $O/ab.txt
grep "^pace" $O/err.log
