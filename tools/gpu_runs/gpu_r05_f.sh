# round 5: three frames in flight per wavefront (groups of 6) against two (groups of 8), matrix-pipe forms; parity first
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_spectral.py -x -q -m gpu -k "matrix_pipe or automatic" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
ROUNDS=3 timeout 600 python tools/fused_ab.py fused_mel=1,fft_exact=1 fused_mel=2,fft_exact=1,fused_nf=2 fused_mel=3,fft_exact=1,fused_nf=2 fused_mel=2,fft_exact=1,fused_nf=3 fused_mel=3,fft_exact=1,fused_nf=3 fused_mel=3,fft_exact=0,fused_nf=2 fused_mel=3,fft_exact=0,fused_nf=3 2>&1 | grep -E "kernel_ms|vs" > $O/fused_ab.log; cat $O/fused_ab.log
