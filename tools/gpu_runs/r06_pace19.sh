#!/bin/bash
# round 6: sinebuf under the TOLERANT controller (MXG_SINEBUF_PACED=1) against its plan of launches; K2f again (unchanged rule)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export MXG_PRINT_PACE=1
O=gpurun_out/r06pace19; mkdir -p $O; rm -f $O/err.log
for r in 1 2; do
for V in 98304 131072 196608 262144; do
MXG_SINEBUF_PACED=1 timeout 300 python bench.py --voices $V --no-cpu-baseline --no-extras --no-configs --steps 480 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "sinebuf paced V=$V r$r"
timeout 300 python bench.py --voices $V --no-cpu-baseline --no-extras --no-configs --steps 480 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "sinebuf plan V=$V r$r"
done
timeout 300 python bench.py --workload config3 --no-cpu-baseline --steps 512 --warmup 64 --kernel-events off 2>> $O/err.log | python tools/line_fields.py "K2f modeA r$r"
done | tee $O/ab.txt
grep "^pace" $O/err.log | tail -12
