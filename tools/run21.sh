cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03q
for var in 4 0; do for split in 1 2 3 4; do
  timeout 300 python bench.py --mixdown fused --no-cpu-baseline --no-extras --steps 200 --warmup 20 --tune osc_mix_var=$var --tune osc_mix_split=$split > gpurun_out/r03q/b_v${var}_s${split}.json 2> gpurun_out/r03q/b_v${var}_s${split}.err
done; done
