# round 5: where the fused K1t loses what it gains: the renderer in two rounds of 8 samples without the marks wavefronts (A/B build pc8),
# the marks wavefronts on the three-instruction chain (A/B build form1)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O
for r in 1 2; do
  for cfg in "lib:1" "lib:0" "ab_pc8:0" "ab_form1:1"; do
    lib=${cfg%%:*}; f=${cfg##*:}
    if [ $lib = lib ]; then unset MXG_LIB; else export MXG_LIB=$R/maximilian_amd/$lib.so; fi
    timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 200 --warmup 20 --tune tab_fused=$f 2>/dev/null | python tools/line_fields.py "tables $lib tab_fused=$f r$r"
  done
done | tee $O/bench2.log
