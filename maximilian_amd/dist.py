"""Multi-GPU host logic: one process per GPU over RCCL / xGMI.

Voice banks, grain streams and FFT frames shard embarrassingly: rank g owns the contiguous global units
[g*per_rank, (g+1)*per_rank) with private state, tables are replicated, and the per-unit render needs no
collective.  The only exchange step on the path is the maxiMix mixdown: every rank reduces its shard to a
[samples, channels] fp64 block on its own GPU and the root receives the sum.

The product's exchange lives in libmaxigpu.so (csrc/comm.hip): `mxg_comm_create` (ncclCommInitRank) and the batching
queue `mxg_mixq_*` -- the local mixes of M consecutive blocks are staged and ONE ncclReduce per M blocks runs on the
queue's own stream while the caller's stream renders the next batch.  This module is the host glue:

  * `create_comm`   rank 0 draws the ncclUniqueId, `torch.distributed` (any backend) carries its 128 bytes to the
                    other ranks, every rank builds its RCCL communicator through the C-ABI;
  * `RcclMixQueue`  ctypes face of mxg_mixq (the product path);
  * `HostMixQueue`  the same slot / push / flush protocol over host buffers and a `torch.distributed` reduce -- used by
                    the CPU tests (gloo, world size 2) to drive the SAME step function without a GPU;
  * `MixdownStep`   one block of the sharded path: render this rank's shard + local mix into the queue's slot, push.
"""
import ctypes

import numpy as np


def shard_range(rank, world, units_per_rank):
    """Global unit indices [lo, hi) owned by `rank` (weak scaling: fixed units per rank)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return rank * units_per_rank, (rank + 1) * units_per_rank


def bank_parameters(lo, hi, total_voices):
    """Synthetic config-2 parameters for global voices [lo, hi): freq = 20 + (v mod 65536)*0.30517578125 Hz
    (exact in binary), pan x_v = v/(total-1)."""
    v = np.arange(lo, hi, dtype=np.float64)
    freq = 20.0 + (v % 65536) * 0.30517578125
    pan = v / max(total_voices - 1, 1)
    return freq, pan


def stream_parameters(lo, hi, total_streams):
    """Synthetic config-5 parameters for global grain streams [lo, hi): start position s/S,
    speed 0.25 + 1.5*(s mod 97)/96, pan x_s = s/(S-1)."""
    s = np.arange(lo, hi, dtype=np.float64)
    pos = s / float(total_streams)
    speed = 0.25 + 1.5 * (s % 97) / 96
    pan = s / max(total_streams - 1, 1)
    return pos, speed, pan


def create_comm(dist, rank, world, device=None):
    """RCCL communicator through the C-ABI.  `dist` is an initialised torch.distributed (used only to carry the
    128-byte id); world == 1 returns None (the queue then degenerates to device copies)."""
    if world <= 1:
        return None
    import torch
    from ._lib import check, lib
    L = lib()
    idbuf = ctypes.create_string_buffer(128)
    if rank == 0:
        check(L.mxg_comm_unique_id(idbuf), "mxg_comm_unique_id")
    t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().numpy().tobytes())
    comm = L.mxg_comm_create(ctypes.c_char_p(raw), world, rank)
    if not comm:
        raise RuntimeError("mxg_comm_create: " + L.mxg_last_error().decode())
    return comm


class RcclMixQueue:
    """mxg_mixq: batched, overlapped sum-reduce of [block_doubles] mix blocks to `root` (include/maxigpu.h)."""

    def __init__(self, comm, block_doubles, depth_blocks=16, root=0, stream=None, groups=1):
        from ._lib import lib
        self.L = lib()
        self.block, self.depth, self.stream, self.groups = int(block_doubles), int(depth_blocks), stream, int(groups)
        # groups > 1: a slot is [groups][block] partial rows (mxg_osc_render_mix_rows) which the queue adds on its own stream
        self.q = self.L.mxg_mixq_create_grouped(comm, self.block, self.depth, root, self.groups)
        if not self.q:
            raise RuntimeError("mxg_mixq_create: " + self.L.mxg_last_error().decode())

    def slot(self):
        p = self.L.mxg_mixq_slot(self.q, self.stream)
        if not p:
            raise RuntimeError("mxg_mixq_slot: " + self.L.mxg_last_error().decode())
        return p

    def push(self):
        from ._lib import check
        check(self.L.mxg_mixq_push(self.q, self.stream), "mxg_mixq_push")

    def flush(self):
        from ._lib import check
        check(self.L.mxg_mixq_flush(self.q, self.stream), "mxg_mixq_flush")

    def result(self):
        """(device pointer, blocks) of the most recently submitted batch's sum (root only)."""
        nb, nbatch = ctypes.c_size_t(0), ctypes.c_size_t(0)
        p = self.L.mxg_mixq_result(self.q, ctypes.byref(nb), ctypes.byref(nbatch))
        return p, nb.value, nbatch.value

    def result_numpy(self):
        """Flush, wait, and download the last batch (tests / offline renders)."""
        from ._lib import check
        self.flush()
        check(self.L.mxg_stream_sync(self.stream), "mxg_stream_sync")
        p, nb, _ = self.result()
        out = np.empty((nb, self.block), np.float64)
        if nb:
            check(self.L.mxg_memcpy_d2h(out.ctypes.data, p, out.nbytes, self.stream), "mxg_memcpy_d2h")
        return out

    def close(self):
        if self.q:
            self.L.mxg_mixq_destroy(self.q)
            self.q = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostMixQueue:
    """The slot / push / flush protocol of mxg_mixq over host buffers and torch.distributed.reduce (gloo in the CPU
    tests).  Same batching, same double buffering, same result semantics -- so the step function the product runs is
    the one the CPU tests exercise."""

    def __init__(self, dist, block_doubles, depth_blocks=16, root=0, groups=1):
        import torch
        self.dist, self.block, self.depth, self.root, self.groups = dist, int(block_doubles), int(depth_blocks), root, int(groups)
        # groups > 1: a slot is [groups][block] partial rows, added when the batch is submitted in the device fold's order (16 interleaved
        # chains, then the chains left to right: mix_partials_kernel) -- the same roundings as mxg_mixq
        self.gstage = [torch.zeros((self.depth, self.groups, self.block), dtype=torch.float64) for _ in range(2)] if self.groups > 1 else None
        self.stage = [torch.zeros((self.depth, self.block), dtype=torch.float64) for _ in range(2)]
        self.res = [torch.zeros((self.depth, self.block), dtype=torch.float64) for _ in range(2)]
        self.work = [None, None]
        self.cur, self.fill, self.last, self.last_blocks, self.batches = 0, 0, -1, 0, 0
        self.slot_out = False

    def _multi(self):
        return self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1

    def slot(self):
        b = self.cur
        if self.fill == 0 and self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        self.slot_out = True
        if self.gstage is not None:
            return self.gstage[b][self.fill]  # a [groups][block] view
        return self.stage[b][self.fill]  # a [block] view the caller fills in place

    def _submit(self):
        import torch
        b = self.cur
        if self.gstage is not None:
            for k in range(self.fill):  # mix_partials_kernel's order (mxg_lanefold.h): 16 interleaved chains g = w, w + 16, ... each
                chains = []            # summed from 0.0 in ascending g, then the chains added left to right
                for w in range(16):
                    c = torch.zeros(self.block, dtype=torch.float64)
                    for g in range(w, self.groups, 16):
                        c += self.gstage[b][k][g]
                    chains.append(c)
                acc = chains[0]
                for c in chains[1:]:
                    acc = acc + c
                self.stage[b][k].copy_(acc)
        self.res[b][:self.fill].copy_(self.stage[b][:self.fill])
        if self._multi():
            self.work[b] = self.dist.reduce(self.res[b][:self.fill], dst=self.root, op=self.dist.ReduceOp.SUM, async_op=True)
        self.last, self.last_blocks = b, self.fill
        self.batches += 1
        self.cur ^= 1
        self.fill = 0

    def push(self):
        assert self.slot_out, "push without slot"
        self.slot_out = False
        self.fill += 1
        if self.fill == self.depth:
            self._submit()

    def flush(self):
        assert not self.slot_out
        if self.fill:
            self._submit()
        for b in range(2):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None

    def result_numpy(self):
        self.flush()
        if self.last < 0:
            return np.empty((0, self.block))
        return self.res[self.last][:self.last_blocks].numpy().copy()


class TorchMixQueue:
    """FALLBACK exchange for bench.py only: the slot / push / flush protocol of mxg_mixq with torch DEVICE tensors as staging and
    torch.distributed.reduce (RCCL through torch's own communicator) as the sum -- used when the library's communicator cannot be
    created on a multi-GPU node (mxg_comm_create failing), so that the scaling run still measures a real cross-GPU reduce and says
    so in its line.  Same batching, same double buffering; the per-workgroup rows of a grouped slot are added by the library's own
    kernel (mxg_mix_rows_sum) on the launch stream.  The launch stream must be torch's current stream."""

    def __init__(self, dist, block_doubles, depth_blocks=16, root=0, stream=None, device=None, groups=1):
        import torch
        from ._lib import lib
        self.L, self.dist = lib(), dist
        self.block, self.depth, self.root, self.stream, self.groups = int(block_doubles), int(depth_blocks), root, stream, int(groups)
        kw = dict(dtype=torch.float64, device=device)
        self.gstage = [torch.zeros((self.depth, self.groups, self.block), **kw) for _ in range(2)] if self.groups > 1 else None
        self.stage = [torch.zeros((self.depth, self.block), **kw) for _ in range(2)]
        self.work = [None, None]
        self.cur, self.fill, self.last, self.last_blocks, self.batches = 0, 0, -1, 0, 0
        self.slot_out = False

    def slot_tensor(self):
        b = self.cur
        if self.fill == 0 and self.work[b] is not None:
            self.work[b].wait()  # (the current stream waits for the reduce that last read this buffer; the host does not)
            self.work[b] = None
        self.slot_out = True
        return self.gstage[b][self.fill] if self.gstage is not None else self.stage[b][self.fill]

    def slot(self):
        return self.slot_tensor().data_ptr()

    def _submit(self):
        from ._lib import check
        b = self.cur
        if self.gstage is not None:
            for k in range(self.fill):
                check(self.L.mxg_mix_rows_sum(self.groups, self.block, self.gstage[b][k].data_ptr(), self.stage[b][k].data_ptr(),
                                              self.stream), "mxg_mix_rows_sum")
        self.work[b] = self.dist.reduce(self.stage[b][:self.fill], dst=self.root, op=self.dist.ReduceOp.SUM, async_op=True)
        self.last, self.last_blocks = b, self.fill
        self.batches += 1
        self.cur ^= 1
        self.fill = 0

    def push(self):
        assert self.slot_out, "push without slot"
        self.slot_out = False
        self.fill += 1
        if self.fill == self.depth:
            self._submit()

    def flush(self):
        assert not self.slot_out
        if self.fill:
            self._submit()
        for b in range(2):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None

    def result_numpy(self):
        import torch
        self.flush()
        if self.stage[0].is_cuda:
            torch.cuda.synchronize()
        if self.last < 0:
            return np.empty((0, self.block))
        return self.stage[self.last][:self.last_blocks].cpu().numpy().copy()

    def close(self):
        pass


class MixdownStep:
    """One block of the sharded path.  `render_mix(slot)` enqueues this rank's render + local maxiMix mixdown of one
    block, writing the [block_doubles] mix at `slot` (a device pointer for RcclMixQueue, a host tensor view for
    HostMixQueue); the queue batches the blocks and reduces them to the root."""

    def __init__(self, render_mix, queue):
        self.render_mix, self.queue = render_mix, queue
        self.blocks = 0

    def __call__(self):
        slot = self.queue.slot()
        self.render_mix(slot)
        self.queue.push()
        self.blocks += 1

    def finish(self):
        self.queue.flush()
