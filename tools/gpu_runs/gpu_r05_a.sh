# round 5, first GPU call: parity of the matrix-pipe forms of the fused FFT+MFCC kernel (fused_mel 2 / 3) and of K1t's new marks
# pass + side fold; then what decides their defaults: the six fused forms interleaved, K1t old / new, K1m producer tick flavours;
# then counters: the fused kernel (vector form and matrix form), playAtSpeed.
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_osctab.py -x -q -m gpu -k "fused or matrix_pipe or osctab or tables" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python tools/fused_ab.py > $O/fused_ab.log 2>&1; tail -14 $O/fused_ab.log
for round in 1 2; do
  for lib in maximilian_amd/libmaxigpu.so maximilian_amd/ab_tabr4.so; do
    MXG_LIB=$R/$lib timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python tools/line_fields.py "tables $lib r$round"
  done
  for lib in maximilian_amd/libmaxigpu.so maximilian_amd/ab_pcfl0.so maximilian_amd/ab_pcfl0x.so; do
    MXG_LIB=$R/$lib timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --no-configs --steps 300 --warmup 30 2>/dev/null | python tools/line_fields.py "k1m $lib r$round"
  done
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --steps 300 --warmup 30 2>/dev/null | python tools/line_fields.py "k1 r$round"
done > $O/ab.log 2>&1; cat $O/ab.log
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" \
           "SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS"; do
  i=$((i+1))
  for only in 0 2; do
    ONLY=$only REPS=3 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_fused/form$only/g$i -o k -- python $R/tools/fused_ab.py > $O/pmc_fused.form$only.g$i.log 2>&1
  done
done
for only in 0 2; do
  ONLY=$only REPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fused/form$only -o k -- python $R/tools/fused_ab.py > $O/trace_fused.form$only.log 2>&1
done
cd $R
bash tools/pmc_sq.sh r05a speedplayer > $O/pmc_sp.log 2>&1
bash tools/pmc_mem.sh r05a speedplayer >> $O/pmc_sp.log 2>&1
ls $R/gpurun_out/ | head -30
