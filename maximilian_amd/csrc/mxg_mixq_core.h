// mxg_mixq_core.h -- the protocol of the batched mix queue (mxg_mixq, include/maxigpu.h), written once against a small
// device interface so that the SAME text runs in the product (HIP streams / events + ncclReduce, comm.hip) and in the CPU
// suite (tests/host_mixq.cpp: streams are worker threads that execute their operations asynchronously with random delays,
// events are real synchronisation objects, the reduce is a two-rank rendezvous) -- SURVEY.md 8(e): one exchange step on the
// path, the sum of the per-rank [samples][channels] maxiMix blocks onto the root.
//
// Protocol.  Two staging buffers [depth][block].  The caller's stream fills slot after slot of staging buffer `cur`
// (slot -> enqueue the local mix into it -> push); when `depth` blocks are in, submit():
//     filled[b]   recorded on the caller's stream            the batch is complete once this event completes
//     qstream waits filled[b]; qstream waits consumed[b]     (if the consumer declared reads of result[b], see release())
//     reduce(stage[b] -> result[b]) on qstream                ONE reduce per batch
//     [root, optional] result[b] -> pinned host ring, block by block
//     reduced[b]  recorded on qstream
// and the caller goes on with the OTHER staging buffer while the reduce runs.  Grouped slots (groups > 1, round 4): the caller's
// kernel leaves `groups` partial rows per block (K1m: one per workgroup of 256 voices) in gstage[b] and the QUEUE adds them, for the
// whole batch at once, on qstream in front of the reduce (fold) -- the per-block "sum the workgroup rows" kernel leaves the render
// stream.  Before the first slot of a buffer is handed out
// again the caller's stream waits reduced[b] (the reduce that last READ this staging buffer).  flush() submits a partial batch
// and makes the caller's stream wait for every outstanding reduce.  Nothing here blocks the host.
#pragma once
#include <stddef.h>

namespace mxg {

// Dev must provide:
//   typedef Stream, Event
//   int record(Event, Stream);  int wait(Stream, Event);
//   int reduce(const double *send, double *recv, size_t count, int root, Stream);   // sum over ranks onto root
//   int copy_to_host(double *h_dst, const double *d_src, size_t count, Stream);
//   int fold(const double *src, double *dst, size_t blocks, size_t groups, size_t block, Stream);   // dst[k][i] = sum_g src[k][g][i], g ascending
//   bool is_root(int root);
template <class Dev>
struct MixQueueCore {
    typedef typename Dev::Stream Stream;
    typedef typename Dev::Event Event;

    Dev *dev = nullptr;
    size_t block = 0;  // doubles per block (samples * channels)
    int depth = 1;     // M blocks per reduce
    int root = 0;
    double *stage[2] = {nullptr, nullptr};   // [M][block] local mixes
    double *result[2] = {nullptr, nullptr};  // [M][block] reduced (meaningful on the root)
    size_t groups = 1;                       // partial rows per block the caller's kernel writes (1: it writes the mix itself)
    double *gstage[2] = {nullptr, nullptr};  // [M][groups][block] (groups > 1 only): what slot() hands out
    Event filled[2] = {}, reduced[2] = {}, consumed[2] = {};
    bool in_flight[2] = {false, false};      // a reduce has been enqueued that reads stage[b]
    bool has_consumer[2] = {false, false};   // release() recorded reads of result[b] that the next reduce into it must wait for
    Stream qstream = {};
    int cur = 0;   // staging buffer being filled
    int fill = 0;  // blocks pushed into it
    bool slot_out = false;
    int last = -1;  // buffer of the most recently submitted batch
    size_t last_blocks = 0;
    size_t batches = 0;
    double *h_sink = nullptr;  // optional pinned host ring [sink_blocks][block] the root copies every batch into
    size_t sink_blocks = 0, sink_pos = 0;

    int submit(Stream caller) {
        const int b = cur;
        const size_t count = (size_t)fill * block;
        if (int s = dev->record(filled[b], caller)) return s;
        if (int s = dev->wait(qstream, filled[b])) return s;
        if (has_consumer[b]) {  // result[b] is about to be overwritten: the declared reads of the batch before last come first
#ifndef MXG_MIXQ_MUTATE_NO_CONSUMED_WAIT
            if (int s = dev->wait(qstream, consumed[b])) return s;
#endif
            has_consumer[b] = false;
        }
        if (groups > 1)
            if (int s = dev->fold(gstage[b], stage[b], (size_t)fill, groups, block, qstream)) return s;
        if (int s = dev->reduce(stage[b], result[b], count, root, qstream)) return s;
        if (h_sink && dev->is_root(root)) {
            for (int i = 0; i < fill; i++) {
                if (int s = dev->copy_to_host(h_sink + (sink_pos % sink_blocks) * block, result[b] + (size_t)i * block, block, qstream))
                    return s;
                sink_pos++;
            }
        }
        if (int s = dev->record(reduced[b], qstream)) return s;
        in_flight[b] = true;
        last = b;
        last_blocks = (size_t)fill;
        batches++;
        cur ^= 1;
        fill = 0;
        return 0;
    }

    // device pointer of the current block's slot; *status != 0 on failure
    double *slot(Stream st, int *status) {
        const int b = cur;
        *status = 0;
        if (fill == 0 && in_flight[b]) {
            // the reduce (and, with groups, the fold in front of it) that last read this staging buffer must be done before the
            // caller's stream overwrites it
#ifndef MXG_MIXQ_MUTATE_NO_SLOT_WAIT  // (tests/host_mixq.cpp builds a mutant without this wait and must catch it)
            if ((*status = dev->wait(st, reduced[b]))) return nullptr;
#endif
            in_flight[b] = false;
        }
        slot_out = true;
        return groups > 1 ? gstage[b] + (size_t)fill * groups * block : stage[b] + (size_t)fill * block;
    }

    int push(Stream st) {
        slot_out = false;
        fill++;
        if (fill < depth) return 0;
        return submit(st);
    }

    int flush(Stream st) {
        if (fill > 0)
            if (int s = submit(st)) return s;
        for (int b = 0; b < 2; b++)
            if (in_flight[b])
                if (int s = dev->wait(st, reduced[b])) return s;  // in_flight stays set: slot() re-waits, harmless
        return 0;
    }

    // the consumer has enqueued, on `st`, every read of the most recent result it is going to make
    int release(Stream st) {
        if (last < 0) return 0;
        if (int s = dev->record(consumed[last], st)) return s;
        has_consumer[last] = true;
        return 0;
    }
};

}  // namespace mxg
