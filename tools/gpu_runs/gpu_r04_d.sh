#!/bin/bash
# round 4, fourth GPU call: A/B of chunked ticks and the store asm's memory clobber (K1 and K1m), SQ counters of K1 / K1m
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
LIBS="maximilian_amd/libmaxigpu.so maximilian_amd/ab_chunk_noclob.so maximilian_amd/ab_nochunk_noclob.so maximilian_amd/ab_nochunk_clob.so maximilian_amd/ab_k1mchunk_clob.so"
for round in 1 2 3; do for lib in $LIBS; do
  for mode in "k1 --no-extras" "k1m --mixdown fused" "k1m_mixonly --mixdown fused --mix-only" "sb4 --no-extras --waveform sinebuf4"; do
    set -- $mode; name=$1; shift
    MXG_LIB=$R/$lib timeout 300 python bench.py --no-cpu-baseline --steps 600 --warmup 50 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', '$lib', 'round $round', 'step_ms', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))
" >> $O/ab.txt
  done
done; done
sort $O/ab.txt | awk '{k=$1" "$2; s[k]=s[k]" "$6} END{for(k in s) print k, s[k]}' | sort
cd /tmp && export TMPDIR=/tmp
for cfg in "k1 --no-extras" "k1m --mixdown fused" "k1m_mixonly --mixdown fused --mix-only"; do
  set -- $cfg; name=$1; shift
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/sq_$name/g$i -o k -- python $R/bench.py --no-cpu-baseline --kernel-events off --steps 10 --warmup 2 "$@" > $O/sq_$name.g$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for name in ("k1", "k1m", "k1m_mixonly"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/r04d/sq_%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            if "osc_" in r["Kernel_Name"]:
                k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    print("##", name)
    for k in sorted(acc):
        v = acc[k][0] / max(acc[k][1], 1)
        print("%-26s %.4g per launch, %.2f per voice-sample" % (k, v, v / (65536 * 512 / 64.0)))
PY
