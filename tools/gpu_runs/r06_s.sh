#!/bin/bash
# round 6, call s: shader-core and L2 counters of the HBM-resident sample bank's kernel (separate rocprofv3 --pmc passes, no trace domain)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sq_r06sb; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  REPS=2 rocprofv3 --pmc $grp --output-format csv -d $OUT/samplebank/g$i -o k -- python $R/tools/bench_samplebank.py > $OUT/samplebank.g$i.log 2>&1
done
find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*.db" -delete
cd $R; tail -2 $OUT/samplebank.g1.log; ls $OUT/samplebank
