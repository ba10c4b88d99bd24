# round 5: the matrix forms with ONE frame in flight and ten wavefronts per CU (fused_layout 2) against two frames x eight
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
ROUNDS=3 timeout 600 python tools/fused_ab.py fused_mel=3,fft_exact=1,fused_layout=0 fused_mel=3,fft_exact=1,fused_layout=2 fused_mel=3,fft_exact=0,fused_layout=0 fused_mel=3,fft_exact=0,fused_layout=2 2>&1 | grep -E "kernel_ms|vs|rror" > $O/fused_ab.log; cat $O/fused_ab.log
