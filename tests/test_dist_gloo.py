"""CPU test (-m "not gpu"): the N>1 path's host logic with world_size 2 over gloo.

Each rank takes its voice shard (maximilian_amd.dist.shard_range / bank_parameters), renders it with
the CPU oracle standing in for the GPU kernels (the product's kernels need a GPU; the sharding,
double-buffered reducer and the reduce-to-rank-0 are the code under test), mixes it to stereo and
hands the [B,2] block to MixReducer.  Rank 0 must end up with the mix of the WHOLE bank."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from maximilian_amd.dist import MixReducer, bank_parameters, shard_range
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = pyoracle.port()
    orc.settings(44100, 2, 1024)
    Vr, B, blocks = 96, 64, 3
    lo, hi = shard_range(rank, world, Vr)
    freq, pan = bank_parameters(lo, hi, Vr * world)
    red = MixReducer(dist, lambda: torch.zeros((B, 2), dtype=torch.float64))
    phase = hold = None
    got = []
    for _ in range(blocks):
        out, phase, hold = orc.osc(8, freq, B, phase=phase, hold=hold)
        buf = red.next_buffer()
        buf.copy_(torch.from_numpy(orc.mix_stereo(out, pan)))
        res = red.submit()
        red.drain()
        got.append(res.clone().numpy())
    if rank == 0:
        q.put(np.stack(got))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_helpers():
    from maximilian_amd.dist import bank_parameters, shard_range
    assert shard_range(0, 8, 65536) == (0, 65536) and shard_range(7, 8, 65536) == (7 * 65536, 8 * 65536)
    with pytest.raises(ValueError):
        shard_range(8, 8, 4)
    f, p = bank_parameters(65536, 65540, 2 * 65536)
    assert f[0] == 20.0 and f[1] == 20.0 + 0.30517578125   # the frequency pattern repeats every 65 536 voices
    assert p[0] == 65536 / (2 * 65536 - 1.0)
    # shards tile the bank without gaps or overlap
    edges = [shard_range(r, 4, 10) for r in range(4)]
    assert [e[0] for e in edges[1:]] == [e[1] for e in edges[:-1]]


def test_two_rank_mix_reduce_gloo(port):
    import torch.multiprocessing as mp
    from maximilian_amd.dist import bank_parameters
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    prt = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, prt, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # expected: the whole 192-voice bank on one "device", sequential reference sum
    Vr, B, blocks = 96, 64, 3
    freq, pan = bank_parameters(0, 2 * Vr, 2 * Vr)
    out, _, _ = port.osc(8, freq, B * blocks)
    exp = port.mix_stereo(out, pan).reshape(blocks, B, 2)
    assert np.abs(got - exp).max() <= 1e-12 * 2 * Vr   # cross-rank sum order != sequential order
    assert np.abs(got).max() > 0.1
