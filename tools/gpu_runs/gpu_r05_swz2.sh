# round 5, end: XOR-folded X image (shipped build) against pad8 (ab_pad8.so), three rounds per process (the first one warms the clocks), processes alternating
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05swz; mkdir -p $O
for r in 1 2 3; do
  for lib in libmaxigpu ab_pad8; do
    MXG_LIB=$R/maximilian_amd/$lib.so ROUNDS=3 timeout 300 python tools/fused_ab.py fused_mel=3,fft_exact=1 fused_mel=2,fft_exact=1 fused_mel=1,fft_exact=1 fused_mel=3,fft_exact=0 2>&1 | grep kernel_ms | sed "s/^/$lib r$r /"
  done
done | tee $O/ab2.log
