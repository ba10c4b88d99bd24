# round 5, end: the shipped fused FFT + MFCC kernel (matrix form, LDS access merging off) under the SQ counters once more:
# what is left of the LDS bank conflicts, where the wavefronts wait
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_INST_ANY SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED"; do
  i=$((i+1))
  ONLY=0 REPS=3 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc/form3/g$i -o k -- python $R/tools/fused_ab.py fused_mel=3,fft_exact=1 > $O/pmc.g$i.log 2>&1
done
python $R/tools/pmc_condense.py $O/pmc $O/pmc_fused_final.json fft_mfcc; rm -rf $O/pmc
python - <<'P'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05pmc2/pmc_fused_final.json'))
for tag,ks in d['counters'].items():
    for k,cs in ks.items():
        for c,v in sorted(cs.items()):
            print(tag,k[-30:],c,'%.2f per frame'%(v['mean']/1048576))
P
