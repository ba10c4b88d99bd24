# round 5: counters of the fused kernel (vector form = setting 0, matrix form = setting 2), playAtSpeed counters, the marks pass of K1t
# across bank sizes (old / new chain), parity of the sampler / drop-in changes
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_dropin.py tests/test_gpu_spectral.py -x -q -m gpu -k "sampler or public_members or matrix_pipe" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for V in 32768 65536 131072 262144; do for lib in maximilian_amd/libmaxigpu.so maximilian_amd/ab_tabr4.so; do
  MXG_LIB=$R/$lib timeout 300 python bench.py --workload tables --voices $V --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python tools/line_fields.py "tables V=$V $lib"
done; done > $O/marks.log 2>&1; cat $O/marks.log
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" \
           "SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for only in 0 2; do
    ONLY=$only REPS=3 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_fused/form$only/g$i -o k -- python $R/tools/fused_ab.py > $O/pmc_fused.form$only.g$i.log 2>&1
    tail -2 $O/pmc_fused.form$only.g$i.log
  done
done
for only in 0 2; do
  ONLY=$only REPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused/form$only -o k -- python $R/tools/fused_ab.py > $O/trace_fused.form$only.log 2>&1
done
cd $R
bash tools/pmc_sq.sh r05b speedplayer > $O/pmc_sp.log 2>&1
bash tools/pmc_mem.sh r05b speedplayer >> $O/pmc_sp.log 2>&1
find $R/gpurun_out -name "*.csv" | head -40; du -sh $R/gpurun_out
