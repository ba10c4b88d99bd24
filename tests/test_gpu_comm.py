"""GPU tests (-m gpu) of the exchange step inside the library (csrc/comm.hip): the batched mix queue
mxg_mixq_* with and without a communicator, and RCCL itself through a ONE-rank communicator (a 1-GPU box cannot host two
ranks -- RCCL refuses two ranks on one device -- so this is as far as a single box goes: librccl resolved by dlopen,
ncclGetUniqueId / ncclCommInitRank / ncclReduce / ncclAllReduce executed on the device).  World size 2 is covered on the
CPU with the same step function over gloo (tests/test_dist_gloo.py); the driver's N = 2/4/8 run is the xGMI measurement."""
import ctypes

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

V, B = 4096, 512


def _bank_mixes(mx, nblocks, queue=None):
    """nblocks carried blocks of the config-2 bank + fused mixdown; into the queue's slots when given."""
    from maximilian_amd.dist import bank_parameters, MixdownStep
    freq, pan = bank_parameters(0, V, V)
    bank = mx.maxiOscBank(V)
    if queue is None:
        return [bank.render_mix("sinebuf", freq, pan, B, store=False)[1].numpy() for _ in range(nblocks)]
    step = MixdownStep(lambda slot: bank.render_mix("sinebuf", freq, pan, B, store=False, mix=slot), queue)
    for _ in range(nblocks):
        step()
    return step


@pytest.fixture(scope="module")
def comm1(mx):
    L = mx.lib()
    idbuf = ctypes.create_string_buffer(128)
    mx._lib.check(L.mxg_comm_unique_id(idbuf), "mxg_comm_unique_id")
    assert any(idbuf.raw), "ncclGetUniqueId left the id empty"
    c = L.mxg_comm_create(idbuf, 1, 0)
    assert c, L.mxg_last_error().decode()
    assert L.mxg_comm_rank(c) == 0 and L.mxg_comm_size(c) == 1
    yield c
    mx._lib.check(L.mxg_comm_destroy(c), "mxg_comm_destroy")


@pytest.mark.parametrize("with_comm", [False, True])
@pytest.mark.parametrize("depth,nblocks", [(16, 16), (16, 37), (4, 3), (1, 5)])
def test_mix_queue_batches(mx, comm1, with_comm, depth, nblocks):
    """Full batches, a ragged tail reduced by flush, and depth 1: every block's mix comes back bit-identical to the
    un-queued mixdown (one rank: the reduce is the identity), in order."""
    from maximilian_amd.dist import RcclMixQueue
    expect = _bank_mixes(mx, nblocks)
    q = RcclMixQueue(comm1 if with_comm else None, B * 2, depth)
    got, seen = [], 0
    from maximilian_amd.dist import bank_parameters
    freq, pan = bank_parameters(0, V, V)
    bank = mx.maxiOscBank(V)
    for k in range(nblocks):
        slot = q.slot()
        bank.render_mix("sinebuf", freq, pan, B, store=False, mix=slot)
        q.push()
        if (k + 1) % depth == 0:                      # a batch was just submitted: read it before it is overwritten
            r = q.result_numpy()
            assert r.shape == (depth, B * 2)
            got.extend(r)
    if nblocks % depth:
        r = q.result_numpy()                          # flushes the partial batch
        assert r.shape == (nblocks % depth, B * 2)
        got.extend(r)
    _, _, batches = q.result()
    assert batches == (nblocks + depth - 1) // depth
    q.close()
    assert len(got) == nblocks
    for k in range(nblocks):
        assert_bits_equal(got[k].reshape(B, 2), expect[k], "block %d" % k)


@pytest.mark.parametrize("with_comm", [False, True])
@pytest.mark.parametrize("depth,nblocks", [(16, 37), (4, 3), (1, 2)])
def test_grouped_mix_queue(mx, comm1, with_comm, depth, nblocks):
    """mxg_mixq_create_grouped + mxg_osc_render_mix_rows (the config-2 step of bench.py --gpus N): the render leaves the
    per-workgroup rows in the queue's slot and the QUEUE adds them, batch by batch, on its own stream -- the results are the bits
    of the un-queued mxg_osc_render_mix (same rows, same order of additions)."""
    from maximilian_amd.dist import RcclMixQueue, bank_parameters
    L = mx.lib()
    expect = _bank_mixes(mx, nblocks)
    G = L.mxg_osc_mix_groups(V)
    q = RcclMixQueue(comm1 if with_comm else None, B * 2, depth, groups=G)
    freq, pan = bank_parameters(0, V, V)
    bank = mx.maxiOscBank(V)
    got = []
    for k in range(nblocks):
        bank.render_mix("sinebuf", freq, pan, B, store=False, rows=q.slot())
        q.push()
        if (k + 1) % depth == 0:
            got.extend(q.result_numpy())
    if nblocks % depth:
        got.extend(q.result_numpy())
    q.close()
    assert len(got) == nblocks
    for k in range(nblocks):
        assert_bits_equal(got[k].reshape(B, 2), expect[k], "block %d" % k)
    assert not L.mxg_mixq_create_grouped(None, 64, 2, 0, 0)


def test_queue_protocol_errors(mx):
    from maximilian_amd.dist import RcclMixQueue
    L = mx.lib()
    q = RcclMixQueue(None, 64, 2)
    assert L.mxg_mixq_push(q.q, None) < 0             # push without a slot
    assert b"slot" in L.mxg_last_error()
    p = q.slot()
    assert L.mxg_mixq_slot(q.q, None) == p            # asking again before the push hands out the same slot
    assert L.mxg_mixq_flush(q.q, None) < 0            # a slot is still open
    q.push()
    q.close()
    assert not L.mxg_mixq_create(None, 0, 2, 0)
    assert not L.mxg_mixq_create(None, 64, 0, 0)
    assert not L.mxg_mixq_create(None, 64, 2, 1)      # root outside a one-rank world


@pytest.mark.parametrize("all_ranks", [0, 1])
def test_comm_reduce_one_rank(mx, comm1, all_ranks):
    """ncclReduce / ncclAllReduce of fp64 through the C-ABI on a one-rank communicator: out of place and in place."""
    L = mx.lib()
    x = np.random.default_rng(3).standard_normal(B * 2)
    send = mx.DeviceBuffer.from_numpy(x)
    recv = mx.DeviceBuffer((B * 2,))
    mx._lib.check(L.mxg_comm_reduce(comm1, send.ptr, recv.ptr, x.size, 0, all_ranks, None), "mxg_comm_reduce")
    assert_bits_equal(recv.numpy(), x)
    mx._lib.check(L.mxg_comm_reduce(comm1, send.ptr, send.ptr, x.size, 0, all_ranks, None), "in place")
    assert_bits_equal(send.numpy(), x)
    assert L.mxg_comm_reduce(comm1, send.ptr, recv.ptr, x.size, 1, all_ranks, None) < 0   # root outside the communicator


def test_mix_reduce_matches_mix_stereo(mx, comm1):
    """mxg_mix_reduce = maxiMix bus over the rank's voices + the cross-rank sum; one rank: exactly mxg_mix_stereo."""
    L = mx.lib()
    rng = np.random.default_rng(5)
    N, Vv = 130, 1500
    x = rng.uniform(-1, 1, (N, Vv)); pan = rng.uniform(0, 1, Vv)
    dx, dp = mx.DeviceBuffer.from_numpy(x), mx.DeviceBuffer.from_numpy(pan)
    expect = mx.maxiMixBank(Vv).stereo(dx, pan).numpy()
    for comm in (None, comm1):
        mix = mx.DeviceBuffer((N, 2))
        mx._lib.check(L.mxg_mix_reduce(comm, 2, Vv, N, dx.ptr, dp.ptr, None, None, mix.ptr, 0, None), "mxg_mix_reduce")
        assert_bits_equal(mix.numpy(), expect, "mix_reduce")


def test_queue_sink_ring(mx, comm1):
    """mxg_mixq_set_sink: the root also lands every summed block in a pinned host ring, in block order."""
    from maximilian_amd.dist import RcclMixQueue
    L = mx.lib()
    nblocks, depth, ring = 11, 4, 16
    expect = _bank_mixes(mx, nblocks)
    host = L.mxg_host_alloc(ring * B * 2 * 8)
    assert host
    try:
        q = RcclMixQueue(comm1, B * 2, depth)
        mx._lib.check(L.mxg_mixq_set_sink(q.q, host, ring), "mxg_mixq_set_sink")
        _bank_mixes(mx, nblocks, q).finish()
        mx._lib.check(L.mxg_stream_sync(None), "sync")
        q.result_numpy()
        arr = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_double)), shape=(ring, B * 2)).copy()
        q.close()
    finally:
        L.mxg_host_free(host)
    for k in range(nblocks):
        assert_bits_equal(arr[k % ring].reshape(B, 2), expect[k], "sink block %d" % k)
