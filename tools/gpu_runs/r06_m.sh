#!/bin/bash
# round 6, call m: K1t with ONE 256-lane workgroup per CU and three rounds in flight (tab_depth 3) against two workgroups with one each
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06m; mkdir -p $O
MXG_T=1 timeout 600 python - > $O/t_depth.log 2>&1 <<'PY'
import numpy as np, maximilian_amd as mx
L = mx.lib(); mx._lib.check(L.mxg_init(0), "init"); mx.maxiSettings.setup(44100, 2, 1024)
rng = np.random.default_rng(3)
for V, N in ((1000, 512), (4099, 333), (17, 16), (40000, 512), (131072, 512)):
    freq, pan, tabs = rng.uniform(20, 20000, V), rng.uniform(0, 1, V), mx.DeviceBuffer.from_numpy(rng.uniform(-1, 1, (V, 514)))
    res = {}
    for depth in (1, 3):
        L.mxg_tune(b"tab_depth", depth)
        bank = mx.maxiOscBank(V)
        o = [bank.sinebuf_tables(freq, tabs, N, pan=pan, store=V <= 40000) for _ in range(3)]
        res[depth] = ([None if a is None else a.numpy() for a, b in o], [b.numpy() for a, b in o], bank.phase.numpy())
    L.mxg_tune(b"tab_depth", 0)
    for k in range(3):
        if res[1][0][k] is not None:
            assert np.array_equal(res[1][0][k].view(np.uint64), res[3][0][k].view(np.uint64)), (V, N, k)
        assert np.abs(res[1][1][k] - res[3][1][k]).max() < 1e-9, (V, N, k, np.abs(res[1][1][k] - res[3][1][k]).max())
    assert np.array_equal(res[1][2], res[3][2])
    print("depth 3 == depth 1:", V, N)
PY
for r in 1 2; do for d in 1 3; do
timeout 300 python bench.py --workload tables --no-cpu-baseline --steps 208 --warmup 16 --kernel-events pass --verbose --tune tab_depth=$d 2>> $O/err.log | python tools/line_fields.py "tab_depth=$d r$r"
done; done | tee $O/ab.txt
tail -n 6 $O/t_depth.log
