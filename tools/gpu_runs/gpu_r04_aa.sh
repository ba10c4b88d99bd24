#!/bin/bash
# round 4: the read + write bank kernels per rw_store setting (rotated arenas, interleaved rounds), every waveform, the launch shapes of the
# heavy oscillators again after their diet, SQ counters of K1m
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04aa
mkdir -p $O
cd $R
timeout 900 python tools/sweep_rw_store.py --voices 65536,131072 --out $O/rw_store.md > $O/rw_store.log 2>&1; tail -32 $O/rw_store.md
timeout 300 python tools/bench_waveforms.py > $O/waveforms.txt 2>&1; grep -v amdgpu $O/waveforms.txt
timeout 600 python tools/sweep_heavy_osc.py 9 0 > $O/sweep.txt 2>&1; grep -v amdgpu $O/sweep.txt | awk '/^##/{c=0} {c++; if (c<=8) print}'
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/sq_k1m/g$i -o k -- python $R/bench.py --no-cpu-baseline --no-configs --kernel-events off --steps 10 --warmup 2 --mixdown fused > $O/sq_k1m.g$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/r04aa/sq_k1m/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "osc_mix" in r["Kernel_Name"]:
            k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print("## k1m (osc_mixpc_kernel)")
for k in sorted(acc):
    v = acc[k][0] / max(acc[k][1], 1)
    print("%-26s %.4g per launch, %.2f per wavefront-sample" % (k, v, v / (65536 * 512 / 64)))
PY
