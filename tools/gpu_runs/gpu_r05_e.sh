# round 5: the whole spectral + osctab + drop-in test files on the new build (LDS merging off, mel loop pipelined, packed sqrt, automatic
# form; K1t's three-instruction marks step), then the forms timed and K1t's kernels across bank sizes
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_osctab.py tests/test_gpu_fullparity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not config2 and not config3 and not config5 and not large_banks" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
ROUNDS=3 timeout 600 python tools/fused_ab.py fused_mel=1,fft_exact=1 fused_mel=2,fft_exact=1 fused_mel=3,fft_exact=1 fused_mel=3,fft_exact=0 2>&1 | grep -E "kernel_ms|vs" > $O/fused_ab.log; cat $O/fused_ab.log
for lib in maximilian_amd/libmaxigpu.so maximilian_amd/ab_tabr4.so; do echo $lib; MXG_LIB=$R/$lib timeout 300 python tools/bench_osctab_marks.py 65536 131072 262144; done > $O/marks.log 2>&1; cat $O/marks.log
timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python tools/line_fields.py "tables new"
