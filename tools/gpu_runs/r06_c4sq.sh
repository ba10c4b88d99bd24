#!/bin/bash
# round 6: shader-core counters of the shipped fft_mfcc_kernel (config 4, default form), separate --pmc passes of bench.py --workload config4
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_r06c4f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o k -- python $R/bench.py --workload config4 --no-cpu-baseline --no-extras --kernel-events off --steps 4 --warmup 2 > $O/g$i.log 2>&1
done
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; ls $O
