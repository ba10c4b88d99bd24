# round 5: the fused kernel with hipcc's LDS access merging switched off (ds_read2_b64 / ds_write2_b64 -> plain 8-byte accesses), A/B on one box;
# parity of the drop-in patch p5 with the rebuilt host binary; sampler voice counts
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_dropin.py -x -q -m gpu -k "sampler or public_members" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for lib in libmaxigpu.so ab_nolso.so ab_nolso2.so; do
  echo "== $lib"; MXG_LIB=$R/maximilian_amd/$lib ROUNDS=2 timeout 600 python tools/fused_ab.py fused_mel=1,fft_exact=1 fused_mel=2,fft_exact=1 fused_mel=3,fft_exact=1 fused_mel=3,fft_exact=0 2>&1 | grep -E "kernel_ms|vs"
done > $O/fused_lso.log 2>&1; cat $O/fused_lso.log
for lib in ab_nolso.so ab_nolso2.so; do
  MXG_LIB=$R/maximilian_amd/$lib timeout 600 python -m pytest tests/test_gpu_spectral.py -x -q -m gpu -k "fused or matrix_pipe" > $O/pytest_$lib.log 2>&1; tail -2 $O/pytest_$lib.log
done
