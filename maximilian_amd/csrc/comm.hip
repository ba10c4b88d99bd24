// comm.hip -- the one exchange step of the path: the multi-GPU maxiMix mixdown over RCCL / xGMI.
//
// The reference has no multi-device code to cite; the contract is SURVEY.md 8(e): voice banks, grain streams and
// frames shard over the GPUs of a node with private state and NO data-path collective, except that the per-rank
// [samples][channels] fp64 mix blocks (maxiMix::stereo/quad/ambisonic, src/maximilian.cpp:503-541, plus the user-side
// sum over voices, e.g. 15.polysynth/main.cpp:67) are summed onto one GPU:
//     ncclReduce(send = mix_g, recv = mix, count, ncclDouble, ncclSum, root)
// One process per GPU; the host exchanges the 128-byte ncclUniqueId however it likes (torch.distributed, MPI, a
// file) and hands it to mxg_comm_create.
//
// A [512][2] block is 8 KiB: a reduce of that size is latency-bound (tens of us) while a 65 536-voice block renders
// in ~45 us, so the product does not reduce block by block.  mxg_mixq is the batching queue SURVEY 8(e) asks for:
// the local mixes of M consecutive blocks land in one staging buffer ([M][block] doubles, 128 KiB at M = 16), and
// ONE ncclReduce per M blocks runs on the queue's own stream while the caller's stream already renders the next M
// blocks into the second staging buffer.  All ordering is device-side (events): nothing in slot/push blocks the host.
//
// librccl is resolved lazily by SONAME (dlopen "librccl.so.1"): inside a PyTorch process that is the RCCL torch has
// already loaded (one runtime per process), in a plain C++ host it is /opt/rocm/lib's.  libmaxigpu.so itself keeps
// loading on machines without RCCL; the comm entry points then fail loudly (MXG_ERR_INVALID + message).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "mxg_common.h"
#include "mxg_mixq_core.h"
#include "mxg_lanefold.h"  // mix_partials_kernel: the fold of grouped slots

// The RCCL entry points are resolved with dlsym, so only a handful of types are needed at compile time: use the installed
// header where there is one and the same few declarations (ABI of NCCL 2.x / RCCL) where there is not -- libmaxigpu.so builds
// and loads on a ROCm install without the RCCL development files.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;  // ncclFloat64
#endif

namespace mxg {
namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.ok) return MXG_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // the copy this process already has
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MXG_ERR_INVALID, "RCCL is not available: %s", dlerror());
    g_rccl.handle = h;
#define MXG_SYM(field, name)                                                                   \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                                \
    if (!g_rccl.field) return fail(MXG_ERR_INVALID, "librccl has no symbol %s", name)
    MXG_SYM(GetUniqueId, "ncclGetUniqueId");
    MXG_SYM(CommInitRank, "ncclCommInitRank");
    MXG_SYM(CommDestroy, "ncclCommDestroy");
    MXG_SYM(Reduce, "ncclReduce");
    MXG_SYM(AllReduce, "ncclAllReduce");
    MXG_SYM(GetErrorString, "ncclGetErrorString");
#undef MXG_SYM
    g_rccl.ok = true;
    return MXG_OK;
}

int check_nccl(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return MXG_OK;
    return fail(MXG_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}
#define MXG_NCCL(call)                              \
    do {                                            \
        int _s = check_nccl((call), #call);         \
        if (_s) return _s;                          \
    } while (0)

}  // namespace
}  // namespace mxg

struct mxg_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
};

// the device interface of the queue protocol (mxg_mixq_core.h) on HIP + RCCL
struct HipMixDev {
    typedef hipStream_t Stream;
    typedef hipEvent_t Event;
    mxg_comm *comm = nullptr;  // NULL: no communicator, the "reduce" is a device copy
    int record(Event e, Stream s) { return mxg::check_hip(hipEventRecord(e, s), "hipEventRecord"); }
    // An event that has already completed orders nothing any more: no dependency packet goes on the stream.  (The render stream's
    // wait for the reduce that last read a staging buffer -- two batches back, long done -- cost it a ~9 us bubble once per batch:
    // profiles/r05_k1m_gaps.md.)  Not while the stream is being captured: a graph must carry the edge.
    int wait(Stream s, Event e) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone && hipEventQuery(e) == hipSuccess) return MXG_OK;
        (void)hipGetLastError();  // (hipEventQuery's hipErrorNotReady is not an error)
        return mxg::check_hip(hipStreamWaitEvent(s, e, 0), "hipStreamWaitEvent");
    }
    int reduce(const double *send, double *recv, size_t count, int root, Stream s) {
        using namespace mxg;
        if (comm)  // a one-rank communicator still goes through RCCL (the path a 1-GPU box can execute)
            return check_nccl(g_rccl.Reduce(send, recv, count, ncclDouble, ncclSum, root, comm->comm, s), "ncclReduce");
        return check_hip(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    }
    int copy_to_host(double *h, const double *d, size_t count, Stream s) {
        return mxg::check_hip(hipMemcpyAsync(h, d, count * sizeof(double), hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
    }
    // dst[k][i] = sum over the `groups` rows of block k, rows in ascending order within each of 16 interleaved chains, then the
    // chains left to right (mix_partials_kernel): one launch for the whole batch, on the queue's stream
    int fold(const double *src, double *dst, size_t blocks, size_t groups, size_t block, Stream s) {
        using namespace mxg;
        if (blocks == 0) return MXG_OK;
        hipLaunchKernelGGL(mix_partials_kernel, dim3((unsigned)((block + 63) / 64), (unsigned)blocks), dim3(64 * kPartWaves), 0, s,
                           groups, block, src, dst);
        return check_hip(hipGetLastError(), "mix queue fold launch");
    }
    bool is_root(int root) { return !comm || comm->rank == root; }
};

struct mxg_mixq : mxg::MixQueueCore<HipMixDev> {
    HipMixDev hip;
};

using namespace mxg;

extern "C" {

int mxg_comm_unique_id(void *h_id) {
    MXG_REQUIRE(h_id, "null id buffer");
    static_assert(sizeof(ncclUniqueId) == MXG_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (int s = rccl_load()) return s;
    ncclUniqueId id;
    MXG_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(h_id, &id, sizeof(id));
    return MXG_OK;
}

mxg_comm *mxg_comm_create(const void *h_id, int nranks, int rank) {
    if (ensure_init_only()) return nullptr;
    if (!h_id || nranks < 1 || rank < 0 || rank >= nranks) {
        fail(MXG_ERR_INVALID, "mxg_comm_create: bad id / rank %d of %d", rank, nranks);
        return nullptr;
    }
    if (rccl_load()) return nullptr;
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    mxg_comm *c = new mxg_comm;
    c->nranks = nranks;
    c->rank = rank;
    if (check_nccl(g_rccl.CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank")) {
        delete c;
        return nullptr;
    }
    return c;
}

int mxg_comm_destroy(mxg_comm *c) {
    if (!c) return MXG_OK;
    int s = MXG_OK;
    if (c->comm) s = check_nccl(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return s;
}

int mxg_comm_rank(const mxg_comm *c) { return c ? c->rank : 0; }
int mxg_comm_size(const mxg_comm *c) { return c ? c->nranks : 1; }

int mxg_comm_reduce(mxg_comm *c, const double *d_send, double *d_recv, size_t count, int root, int all, void *stream) {
    if (int s = ensure_init()) return s;
    MXG_REQUIRE(d_send && d_recv, "null device pointer");
    hipStream_t st = resolve_stream(stream);
    const int nranks = c ? c->nranks : 1;
    MXG_REQUIRE(root >= 0 && root < nranks, "root outside the communicator");
    if (count == 0) return MXG_OK;
    if (!c) {
        if (d_recv != d_send)
            MXG_HIP(hipMemcpyAsync(d_recv, d_send, count * sizeof(double), hipMemcpyDeviceToDevice, st));
        return MXG_OK;
    }
    if (all)
        MXG_NCCL(g_rccl.AllReduce(d_send, d_recv, count, ncclDouble, ncclSum, c->comm, st));
    else
        MXG_NCCL(g_rccl.Reduce(d_send, d_recv, count, ncclDouble, ncclSum, root, c->comm, st));
    return MXG_OK;
}

int mxg_mix_reduce(mxg_comm *c, int channels, size_t V, size_t N, const double *d_in, const double *d_x,
                   const double *d_y, const double *d_z, double *d_mix, int root, void *stream) {
    // this rank's voices -> [N][channels] (K3), then the sum over ranks lands in the root's d_mix
    if (int s = mxg_mix_bus(channels, V, N, d_in, d_x, d_y, d_z, nullptr, d_mix, stream)) return s;
    return mxg_comm_reduce(c, d_mix, d_mix, N * (size_t)channels, root, 0, stream);
}

mxg_mixq *mxg_mixq_create(mxg_comm *c, size_t block_doubles, int depth_blocks, int root) {
    return mxg_mixq_create_grouped(c, block_doubles, depth_blocks, root, 1);
}

mxg_mixq *mxg_mixq_create_grouped(mxg_comm *c, size_t block_doubles, int depth_blocks, int root, size_t groups) {
    if (ensure_init_only()) return nullptr;
    if (block_doubles == 0 || depth_blocks < 1 || depth_blocks > 4096 || root < 0 || root >= (c ? c->nranks : 1) || groups < 1 ||
        groups > ((size_t)1 << 20)) {
        fail(MXG_ERR_INVALID, "mxg_mixq_create: bad block size / depth / root / groups");
        return nullptr;
    }
    mxg_mixq *q = new mxg_mixq;
    q->hip.comm = c;
    q->dev = &q->hip;
    q->block = block_doubles;
    q->depth = depth_blocks;
    q->root = root;
    q->groups = groups;
    const size_t bytes = block_doubles * (size_t)depth_blocks * sizeof(double);
    bool ok = hipStreamCreateWithFlags(&q->qstream, hipStreamNonBlocking) == hipSuccess;
    for (int b = 0; b < 2 && ok; b++) {
        ok = ok && hipMalloc(&q->stage[b], bytes) == hipSuccess && hipMalloc(&q->result[b], bytes) == hipSuccess;
        ok = ok && hipMemset(q->stage[b], 0, bytes) == hipSuccess && hipMemset(q->result[b], 0, bytes) == hipSuccess;
        if (groups > 1) ok = ok && hipMalloc(&q->gstage[b], bytes * groups) == hipSuccess && hipMemset(q->gstage[b], 0, bytes * groups) == hipSuccess;
        // filled / consumed order streams of THIS device only: no system-scope fence on their record (the render stream records one
        // per batch); reduced also publishes the root's copy to the pinned host ring: it keeps the default fences
        ok = ok && hipEventCreateWithFlags(&q->filled[b], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&q->reduced[b], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&q->consumed[b], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
    }
    if (!ok) {
        fail(MXG_ERR_HIP, "mxg_mixq_create: %s", hipGetErrorString(hipGetLastError()));
        mxg_mixq_destroy(q);
        return nullptr;
    }
    return q;
}

int mxg_mixq_destroy(mxg_mixq *q) {
    if (!q) return MXG_OK;
    if (q->qstream) (void)hipStreamSynchronize(q->qstream);
    for (int b = 0; b < 2; b++) {
        if (q->stage[b]) (void)hipFree(q->stage[b]);
        if (q->result[b]) (void)hipFree(q->result[b]);
        if (q->gstage[b]) (void)hipFree(q->gstage[b]);
        if (q->filled[b]) (void)hipEventDestroy(q->filled[b]);
        if (q->reduced[b]) (void)hipEventDestroy(q->reduced[b]);
        if (q->consumed[b]) (void)hipEventDestroy(q->consumed[b]);
    }
    if (q->qstream) (void)hipStreamDestroy(q->qstream);
    delete q;
    return MXG_OK;
}

int mxg_mixq_set_sink(mxg_mixq *q, double *h_pinned, size_t ring_blocks) {
    MXG_REQUIRE(q, "null queue");
    MXG_REQUIRE((h_pinned && ring_blocks >= (size_t)q->depth) || (!h_pinned && ring_blocks == 0),
                "the host ring must hold at least one batch");
    q->h_sink = h_pinned;
    q->sink_blocks = ring_blocks;
    q->sink_pos = 0;
    return MXG_OK;
}

double *mxg_mixq_slot(mxg_mixq *q, void *stream) {
    if (!q) {
        fail(MXG_ERR_INVALID, "mxg_mixq_slot: null queue");
        return nullptr;
    }
    int status = 0;
    return q->slot(resolve_stream(stream), &status);
}

int mxg_mixq_push(mxg_mixq *q, void *stream) {
    MXG_REQUIRE(q, "null queue");
    MXG_REQUIRE(q->slot_out, "mxg_mixq_push without mxg_mixq_slot");
    return q->push(resolve_stream(stream));
}

int mxg_mixq_flush(mxg_mixq *q, void *stream) {
    MXG_REQUIRE(q, "null queue");
    MXG_REQUIRE(!q->slot_out, "a slot is still open (mxg_mixq_slot without mxg_mixq_push)");
    return q->flush(resolve_stream(stream));
}

int mxg_mixq_release(mxg_mixq *q, void *stream) {
    MXG_REQUIRE(q, "null queue");
    return q->release(resolve_stream(stream));
}

const double *mxg_mixq_result(const mxg_mixq *q, size_t *h_blocks, size_t *h_batches) {
    if (!q || q->last < 0) {
        if (h_blocks) *h_blocks = 0;
        if (h_batches) *h_batches = q ? q->batches : 0;
        return nullptr;
    }
    if (h_blocks) *h_blocks = q->last_blocks;
    if (h_batches) *h_batches = q->batches;
    return q->result[q->last];
}

}  // extern "C"
