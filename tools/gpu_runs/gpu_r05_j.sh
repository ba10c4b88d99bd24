# round 5: where the N > 1 step's distance from the bare K1m kernel sits -- kernel-trace of the fused-mixdown bench, gaps between consecutive
# render kernels by position in the mix queue's 16-block batch; the same for the plain K1 loop
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_k1m -o k -- python $R/bench.py --mixdown fused --no-cpu-baseline --no-extras --no-configs --steps 480 --warmup 32 --kernel-events off > $O/k1m.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_k1 -o k -- python $R/bench.py --no-cpu-baseline --no-extras --no-configs --steps 480 --warmup 32 --kernel-events off > $O/k1.log 2>&1
cd $R
python tools/trace_gaps.py $(find $O/t_k1m -name "*kernel_trace.csv") osc_mix 16 | tee $O/gaps_k1m.txt
python tools/trace_gaps.py $(find $O/t_k1 -name "*kernel_trace.csv") osc_kernel | tee $O/gaps_k1.txt
grep -h "^{" $O/k1m.log | python tools/line_fields.py "k1m profiled"; grep -h "^{" $O/k1.log | python tools/line_fields.py "k1 profiled"
rm -rf $O/t_k1m $O/t_k1
