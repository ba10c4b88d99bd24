#!/bin/bash
# round 6: counters of the workloads whose kernels changed after the first profile pass (K1t one-round form + queue, mode B's polynomial sine, the ring sample bank)
cd $GRAFT_REPO_ROOT
ONLY="config2_tables config3_modB sample_bank" bash tools/profile_r06.sh r06 2>&1 | tail -3
# fp64 flops of the per-sample-modulated voice (the ONLY filter skips the extra passes of the script)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06/config3_modB/pmc_f64 -o b -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --kernel-events off --steps 16 --warmup 2 --workload config3 --voice-mode 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_r06/config3_modB.f64.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_r06 -name "*agent_info.csv" -delete; find $GRAFT_REPO_ROOT/gpurun_out/prof_r06 -name "*.db" -delete
