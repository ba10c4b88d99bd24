# round 5: the mix queue without dependency packets for completed events -- tests of the queue / comm / graph capture, then the fused-mixdown step
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_osc.py -x -q -m gpu -k "comm or mixq or queue or graph or mix" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for r in 1 2 3; do
  timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --no-configs --steps 480 --warmup 32 2>/dev/null | python tools/line_fields.py "k1m r$r"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --steps 480 --warmup 32 2>/dev/null | python tools/line_fields.py "k1 r$r"
done | tee $O/k1m.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_k1m -o k -- python $R/bench.py --mixdown fused --no-cpu-baseline --no-extras --no-configs --steps 480 --warmup 32 --kernel-events off > $O/k1m_prof.log 2>&1
cd $R; python tools/trace_gaps.py $(find $O/t_k1m -name "*kernel_trace.csv") osc_mix 16 | tee $O/gaps_k1m.txt; rm -rf $O/t_k1m
