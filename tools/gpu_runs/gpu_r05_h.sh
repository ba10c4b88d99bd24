# round 5: the row-pitch argument of K1 -- parity, then the sweep over padded pitches; the headline re-measured (K1's kernel gained an argument)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_osc.py -x -q -m gpu -k "pitch or golden or plan" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/sweep_osc_pitch5.py > $O/pitch.log 2>&1; cat $O/pitch.log
for r in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-extras --no-configs --steps 300 --warmup 30 2>/dev/null | python tools/line_fields.py "k1 r$r"; done | tee $O/k1.log
