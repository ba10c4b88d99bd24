# round 5: padded pitches x launch plan for the 2 MB-pitch banks; K1m poll interval / priority A/B
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O
timeout 900 python tools/sweep_osc_pitch5.py 262144 524288 786432 1048576 > $O/pitch.log 2>&1; cat $O/pitch.log
for r in 1 2 3; do
  for lib in libmaxigpu.so ab_cs16.so ab_cs48.so ab_prio3.so ab_prio1.so; do
    MXG_LIB=$R/maximilian_amd/$lib timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --no-configs --steps 300 --warmup 30 2>/dev/null | python tools/line_fields.py "k1m $lib r$r"
  done
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --steps 300 --warmup 30 2>/dev/null | python tools/line_fields.py "k1 r$r"
done | tee $O/k1m.log
