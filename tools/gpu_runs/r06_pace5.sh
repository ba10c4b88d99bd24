#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06pace5; mkdir -p $O
timeout 200 python tools/probes/pace_ctl.py 0 600 2>&1 | awk "NR<30 || NR%25==0" > $O/ctl.txt
tail -25 $O/ctl.txt
for r in 1 2; do for p in 0 1 56; do for d in 2 1; do
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_diet=$d --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA diet=$d r$r"
done
timeout 300 python bench.py --workload config3 --mixdown fused --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeA+mix r$r"
timeout 300 python bench.py --workload config3 --voice-mode 1 --no-cpu-baseline --steps 128 --warmup 128 --kernel-events off --tune voice_pace=$p 2>> $O/err.log | python tools/line_fields.py "pace=$p modeB r$r"
done
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 32 --warmup 8 --kernel-events off --tune voice_diet=2 2>> $O/err.log | python tools/line_fields.py "short run (8 + 32) modeA diet r$r"
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 128 --warmup 16 --kernel-events off --tune voice_diet=2 2>> $O/err.log | python tools/line_fields.py "short run (16 + 128) modeA diet r$r"
done | tee $O/ab.txt
