# round 5: the whole GPU suite, smoke(), the default bench line (with the per-config cpu baselines), fft kernels before / after the LDS-merge switch
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
for lib in ab_fftmerge.so libmaxigpu.so; do MXG_LIB=$R/maximilian_amd/$lib REPS=5 timeout 300 python tools/bench_spectral.py 2>/dev/null | grep -E "fft_mags_ms|fft_mags_phases_ms|fft_plus" | tr '\n' ' '; echo " <- $lib"; done | tee $O/fft_merge.log
