# round 5: parity of the shared-image / NF = 4 forms; the fused forms timed (NF 2 / 4); the marks pass across bank sizes (old / new
# chain); counters of the fused kernel (condensed on the box); playAtSpeed counters
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_dropin.py tests/test_gpu_spectral.py tests/test_gpu_osctab.py -x -q -m gpu -k "sampler or public_members or matrix_pipe or fused or osctab or tables" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
MXG_TUNE_ENV=1 timeout 600 python tools/fused_ab.py fused_mel=1,fft_exact=1 fused_mel=2,fft_exact=1,fused_nf=2 fused_mel=3,fft_exact=1,fused_nf=2 fused_mel=2,fft_exact=1,fused_nf=4 fused_mel=3,fft_exact=1,fused_nf=4 fused_mel=3,fft_exact=0,fused_nf=2 fused_mel=3,fft_exact=0,fused_nf=4 > $O/fused_ab.log 2>&1; tail -16 $O/fused_ab.log
for lib in maximilian_amd/libmaxigpu.so maximilian_amd/ab_tabr4.so; do echo $lib; MXG_LIB=$R/$lib timeout 300 python tools/bench_osctab_marks.py; done > $O/marks.log 2>&1; cat $O/marks.log
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" \
           "SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for only in 0 2 4; do
    ONLY=$only REPS=3 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_fused/form$only/g$i -o k -- python $R/tools/fused_ab.py fused_mel=1,fft_exact=1 x fused_mel=3,fft_exact=1,fused_nf=2 x fused_mel=3,fft_exact=1,fused_nf=4 > $O/pmc_fused.form$only.g$i.log 2>&1
  done
done
for only in 0 2 4; do
  ONLY=$only REPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pmc_fused/form$only/trace -o k -- python $R/tools/fused_ab.py fused_mel=1,fft_exact=1 x fused_mel=3,fft_exact=1,fused_nf=2 x fused_mel=3,fft_exact=1,fused_nf=4 > $O/trace_fused.form$only.log 2>&1
done
python $R/tools/pmc_condense.py $O/pmc_fused $O/pmc_fused.json fft_mfcc; rm -rf $O/pmc_fused
cd $R
bash tools/pmc_sq.sh r05c speedplayer > $O/pmc_sp.log 2>&1
bash tools/pmc_mem.sh r05c speedplayer >> $O/pmc_sp.log 2>&1
python tools/pmc_condense.py $R/gpurun_out/sq_r05c $O/pmc_sp_sq.json sample_; python tools/pmc_condense.py $R/gpurun_out/mem_r05c $O/pmc_sp_mem.json sample_
rm -rf $R/gpurun_out/sq_r05c $R/gpurun_out/mem_r05c
du -sh $R/gpurun_out
