# round 5: config 5's time slices (the first slice's scheduler is the exposed part of the step): 4 (default) / 6 / 8 / 12
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05o; mkdir -p $O
for r in 1 2; do for s in 4 6 8 12; do
  timeout 300 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 --tune grain_slices=$s 2>/dev/null | python tools/line_fields.py "config5 slices=$s r$r"
done; done | tee $O/slices.log
