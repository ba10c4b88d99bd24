# round 5, end: the X image's slot map of the fused FFT + MFCC kernel: XOR fold (shipped build) against pad8 (A/B build ab_pad8.so):
# parity of every fused form, the forms timed interleaved, the LDS conflict counters of the new map
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05swz; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_fullparity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fused or matrix or config4 or mfcc or automatic" 2>&1 | tail -3 | tee $O/tests.log
for r in 1 2 3; do
  for lib in libmaxigpu ab_pad8; do
    MXG_LIB=$R/maximilian_amd/$lib.so ROUNDS=1 timeout 300 python tools/fused_ab.py fused_mel=3,fft_exact=1 fused_mel=2,fft_exact=1 fused_mel=3,fft_exact=0 2>&1 | grep kernel_ms | sed "s/^/$lib r$r /"
  done
done | tee $O/ab.log
cd /tmp && export TMPDIR=/tmp
ONLY=0 REPS=3 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $O/pmc/swz/g1 -o k -- python $R/tools/fused_ab.py fused_mel=3,fft_exact=1 > $O/pmc.log 2>&1
python $R/tools/pmc_condense.py $O/pmc $O/pmc_swz.json fft_mfcc; rm -rf $O/pmc
python - <<'P'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05swz/pmc_swz.json'))
for tag,ks in d['counters'].items():
    for k,cs in ks.items():
        for c,v in sorted(cs.items()):
            print(c,'%.2f per frame'%(v['mean']/1048576))
P
