"""CPU test (-m "not gpu"): the N>1 path's host logic with world_size 2 over gloo.

The step function is the product's (maximilian_amd.dist.MixdownStep: slot -> render + local mix -> push, batched M blocks
per reduce); the kernels need a GPU, so the CPU oracle stands in for them and HostMixQueue (same slot/push/flush protocol as
the C-ABI's mxg_mixq, reduce over gloo) stands in for the RCCL queue.  The BASELINE shapes that shard are driven (round 6: config 3
too -- the subtractive voice bank with carried state, one partial mix row per group of voices, as mxg_voice_render_mix_rows leaves them):
config 2 (voice bank, per-block stereo mixdown, M blocks per reduce incl. a partial last batch) and config 5 (grain
streams, one [T][2] reduce).  Rank 0 must end up with the mix of the WHOLE bank, compared with the oracle's sequential sum."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, mix_tol

CFG2 = dict(Vr=96, B=64, blocks=7, M=3, G=32)    # 7 blocks, 3 per reduce: batches of 3, 3 and a flushed 1; slots of 3 rows of 32 voices
CFG5 = dict(Sr=24, T=1500, L=30000)
CFG3 = dict(Vr=80, B=64, blocks=5, M=2, G=32)    # config 3 sharded (round 6): subtractive voices, carried state, grouped slots of 3 rows


def _voice_params(lo, hi):
    v = np.arange(lo, hi)
    freq = np.minimum(20.0 + v * 7.0, 5000.0)
    return freq, 200 + 4 * freq, 1.0 + (v % 5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grain_sample(L):
    n = np.arange(L)
    rng = np.random.default_rng(0x4D415849)
    return 0.5 * np.sin(2 * np.pi * 110 * n / 44100) + 0.25 * np.sin(2 * np.pi * 331 * n / 44100) + 0.05 * rng.uniform(-1, 1, L)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from maximilian_amd.dist import (HostMixQueue, MixdownStep, bank_parameters, shard_range, stream_parameters)
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = pyoracle.port()
    orc.settings(44100, 2, 1024)

    # ---- config-2 shape: voice shard, block by block, M blocks per reduce --------------------------------
    c = CFG2
    lo, hi = shard_range(rank, world, c["Vr"])
    freq, pan = bank_parameters(lo, hi, c["Vr"] * world)
    # grouped slots, as the product's config-2 step uses them (mxg_osc_render_mix_rows + mxg_mixq_create_grouped): the render
    # leaves one partial mix per group of voices (256 on the device, G here) and the queue adds the rows when it submits a batch
    groups = -(-c["Vr"] // c["G"])
    queue = HostMixQueue(dist, c["B"] * 2, depth_blocks=c["M"], groups=groups)
    state = {"phase": None, "hold": None}
    got2 = []

    def render_mix(slot):
        out, state["phase"], state["hold"] = orc.osc(8, freq, c["B"], phase=state["phase"], hold=state["hold"])
        for g in range(groups):
            sl = slice(g * c["G"], (g + 1) * c["G"])
            slot[g].copy_(torch.from_numpy(orc.mix_stereo(out[:, sl], pan[sl]).reshape(-1)))

    step = MixdownStep(render_mix, queue)
    for k in range(c["blocks"]):
        before = queue.batches
        step()
        if queue.batches != before:  # a full batch was submitted: collect it
            got2.append(queue.result_numpy())
    step.finish()
    if queue.last_blocks and len(got2) * c["M"] < c["blocks"]:
        got2.append(queue.result_numpy())
    got2 = np.concatenate(got2).reshape(-1, c["B"], 2)

    # ---- config-5 shape: grain-stream shard, one [T][2] reduce -------------------------------------------
    g = CFG5
    lo, hi = shard_range(rank, world, g["Sr"])
    pos, speed, pan5 = stream_parameters(lo, hi, g["Sr"] * world)
    smp = _grain_sample(g["L"])
    queue5 = HostMixQueue(dist, g["T"] * 2, depth_blocks=1)

    def render_mix5(slot):
        st = np.zeros((4, hi - lo))
        st[0] = np.clip(pos * g["L"], 0, g["L"] - 1)   # setPosition, L/maxiGrains.h:335-338
        out, _, _, rc = orc.granular(0, 0, smp, g["T"], speed, st=st)
        assert rc == 0
        slot.copy_(torch.from_numpy(orc.mix_stereo(out, pan5).reshape(-1)))

    step5 = MixdownStep(render_mix5, queue5)
    step5()
    step5.finish()
    got5 = queue5.result_numpy().reshape(g["T"], 2)
    # ---- config-3 shape (round 6: mxg_voice_render_mix_rows into a grouped slot): voice shard saw -> lores -> adsr with carried state,
    # one partial mix row per group of voices, M blocks per reduce --------------------------------------------------------------------
    c3 = CFG3
    lo3, hi3 = shard_range(rank, world, c3["Vr"])
    f3, cu3, rs3 = _voice_params(lo3, hi3)
    pan3 = np.arange(lo3, hi3) / (c3["Vr"] * world - 1.0)
    par3 = np.stack([np.full(hi3 - lo3, orc.env_coeff(0, 1)), np.full(hi3 - lo3, orc.env_coeff(1, 5)), np.full(hi3 - lo3, 0.5),
                     np.full(hi3 - lo3, orc.env_coeff(2, 20))])
    hold3 = np.ones(hi3 - lo3, np.int64)
    groups3 = -(-c3["Vr"] // c3["G"])
    queue3 = HostMixQueue(dist, c3["B"] * 2, depth_blocks=c3["M"], groups=groups3)
    vs = {"k": 0, "st": (None, None, None, None)}
    got3 = []

    def render_mix3(slot):
        n0 = vs["k"] * c3["B"]
        trig = ((np.arange(n0, n0 + c3["B"]) % 130) < 70).astype(np.int32)
        out, ost, fst, dst, ist = orc.voice(0, f3, cu3, rs3, trig, par3, hold3, ost=vs["st"][0], fst=vs["st"][1], dstate=vs["st"][2],
                                            istate=vs["st"][3])
        vs["st"] = (ost, fst, dst, ist)
        vs["k"] += 1
        for g in range(groups3):
            sl = slice(g * c3["G"], min((g + 1) * c3["G"], hi3 - lo3))
            slot[g].copy_(torch.from_numpy(orc.mix_stereo(out[:, sl], pan3[sl]).reshape(-1)))

    step3 = MixdownStep(render_mix3, queue3)
    for k in range(c3["blocks"]):
        before = queue3.batches
        step3()
        if queue3.batches != before:
            got3.append(queue3.result_numpy())
    step3.finish()
    if queue3.last_blocks and len(got3) * c3["M"] < c3["blocks"]:
        got3.append(queue3.result_numpy())
    got3 = np.concatenate(got3).reshape(-1, c3["B"], 2)
    # ---- bench.py's fallback exchange (TorchMixQueue: torch tensors as staging, torch.distributed.reduce): the same protocol ------
    from maximilian_amd.dist import TorchMixQueue
    tq = TorchMixQueue(dist, 6, depth_blocks=2, root=0, stream=None, device=None, groups=1)
    got_t = []
    for k in range(5):  # batches of 2, 2 and a flushed 1
        before = tq.batches
        tq.slot_tensor().copy_(torch.full((6,), float((k + 1) * (rank + 1)), dtype=torch.float64) + torch.arange(6, dtype=torch.float64))
        tq.push()
        if tq.batches != before:
            got_t.append(tq.result_numpy())
    tq.flush()
    if len(got_t) * 2 < 5:
        got_t.append(tq.result_numpy())
    got_t = np.concatenate(got_t)
    if rank == 0:
        q.put((got2, got5, got_t, got3))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_helpers():
    from maximilian_amd.dist import bank_parameters, shard_range, stream_parameters
    assert shard_range(0, 8, 65536) == (0, 65536) and shard_range(7, 8, 65536) == (7 * 65536, 8 * 65536)
    with pytest.raises(ValueError):
        shard_range(8, 8, 4)
    f, p = bank_parameters(65536, 65540, 2 * 65536)
    assert f[0] == 20.0 and f[1] == 20.0 + 0.30517578125   # the frequency pattern repeats every 65 536 voices
    assert p[0] == 65536 / (2 * 65536 - 1.0)
    # shards tile the bank without gaps or overlap
    edges = [shard_range(r, 4, 10) for r in range(4)]
    assert [e[0] for e in edges[1:]] == [e[1] for e in edges[:-1]]
    pos, speed, pan = stream_parameters(2048, 2052, 16384)
    assert pos[0] == 2048 / 16384.0 and speed[1] == 0.25 + 1.5 * (2049 % 97) / 96 and pan[0] == 2048 / 16383.0


def test_host_queue_single_process_batches():
    """The queue protocol alone (no process group): batches of M, a flushed partial batch, slots land in order."""
    from maximilian_amd.dist import HostMixQueue, MixdownStep
    q = HostMixQueue(None, 4, depth_blocks=3)
    k = [0]

    def fill(slot):
        slot[:] = float(k[0])
        k[0] += 1
    step = MixdownStep(fill, q)
    seen = []
    for _ in range(8):
        b = q.batches
        step()
        if q.batches != b:
            seen.append(q.result_numpy()[:, 0].tolist())
    step.finish()
    seen.append(q.result_numpy()[:, 0].tolist())
    assert seen == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0], [6.0, 7.0]] and step.blocks == 8


def test_two_rank_mixdown_steps_gloo(port):
    import torch.multiprocessing as mp
    from maximilian_amd.dist import bank_parameters, stream_parameters
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    prt = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, prt, q)) for r in range(2)]
    for p in procs:
        p.start()
    got2, got5, got_t, got3 = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # config-2 shape: the whole 192-voice bank on one "device", the reference's sequential sum
    c = CFG2
    freq, pan = bank_parameters(0, 2 * c["Vr"], 2 * c["Vr"])
    out, _, _ = port.osc(8, freq, c["B"] * c["blocks"])
    exp = port.mix_stereo(out, pan).reshape(c["blocks"], c["B"], 2)
    assert got2.shape == exp.shape
    assert np.abs(got2 - exp).max() <= mix_tol(2 * c["Vr"])   # cross-rank sum order != sequential order
    assert np.abs(got2).max() > 0.1
    # config-5 shape: all 48 streams in one sequential mix
    g = CFG5
    pos, speed, pan5 = stream_parameters(0, 2 * g["Sr"], 2 * g["Sr"])
    st = np.zeros((4, 2 * g["Sr"]))
    st[0] = np.clip(pos * g["L"], 0, g["L"] - 1)
    o5, _, _, rc = port.granular(0, 0, _grain_sample(g["L"]), g["T"], speed, st=st)
    assert rc == 0
    exp5 = port.mix_stereo(o5, pan5)
    assert np.abs(got5 - exp5).max() <= mix_tol(2 * g["Sr"], np.abs(o5).max())
    assert np.abs(got5).max() > 0.05
    # config-3 shape: the whole 160-voice bank on one "device", five carried blocks, the reference's voice-order sum
    c3 = CFG3
    f3, cu3, rs3 = _voice_params(0, 2 * c3["Vr"])
    V3 = 2 * c3["Vr"]
    par3 = np.stack([np.full(V3, port.env_coeff(0, 1)), np.full(V3, port.env_coeff(1, 5)), np.full(V3, 0.5), np.full(V3, port.env_coeff(2, 20))])
    trig3 = ((np.arange(c3["B"] * c3["blocks"]) % 130) < 70).astype(np.int32)
    o3 = port.voice(0, f3, cu3, rs3, trig3, par3, np.ones(V3, np.int64))[0]
    exp3 = port.mix_stereo(o3, np.arange(V3) / (V3 - 1.0)).reshape(c3["blocks"], c3["B"], 2)
    assert got3.shape == exp3.shape
    assert np.abs(got3 - exp3).max() <= mix_tol(V3, np.abs(o3).max(), sums=exp3)
    assert np.abs(got3).max() > 0.05
    # the fallback queue: block k of rank r is (k + 1)(r + 1) + arange(6); the root holds the sum over both ranks
    exp_t = np.stack([3.0 * (k + 1) + 2.0 * np.arange(6) for k in range(5)])
    assert np.array_equal(got_t, exp_t)
