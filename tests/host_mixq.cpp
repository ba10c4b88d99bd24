// host_mixq.cpp -- CPU test harness (tests/test_mixq_host.py): the product's mix-queue protocol (maximilian_amd/csrc/
// mxg_mixq_core.h, the text comm.hip instantiates on HIP + RCCL) driven by TWO ranks with many batches in flight on a
// simulated asynchronous device:
//   * a Stream is a worker thread executing its operations in order, each after a random delay (so anything the protocol does
//     not order explicitly WILL be observed out of order);
//   * an Event has hipEvent semantics (a wait captures the most recent record at the time of the call);
//   * the reduce is a two-rank rendezvous on the queue streams: both ranks deposit their send buffer, the root sums in rank
//     order, nobody leaves before the root has read (as a real ncclReduce may read the send buffer until it completes).
// The "render" of a block is itself an asynchronous operation on the caller's stream that writes the slot when it EXECUTES,
// like a kernel.  Rank r's block k carries value(r, k, i) = (r + 1) * 1000003 + k * 17 + i * 0.5 (exact in fp64); the root's
// results -- read through result() after flush, through release()-ordered asynchronous reads, and through the host sink
// ring -- must equal the sum over ranks for every block of every batch.
// Built a second time with -DMXG_MIXQ_MUTATE_NO_SLOT_WAIT (the wait that protects a staging buffer from being refilled
// while its reduce still reads it is compiled out): the harness must then REPORT corruption, which shows the check has teeth.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "mxg_mixq_core.h"

namespace {

struct FakeEvent {
    std::mutex mu;
    std::condition_variable cv;
    unsigned long recorded = 0, done = 0;
};

struct FakeStream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> ops;
    bool quit = false;
    unsigned seed;
    int max_delay_us;
    std::thread th;
    size_t pending = 0;
    FakeStream(unsigned s, int d) : seed(s), max_delay_us(d), th([this] { run(); }) {}
    ~FakeStream() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        th.join();
    }
    void run() {
        std::minstd_rand rng(seed);
        for (;;) {
            std::function<void()> op;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return quit || !ops.empty(); });
                if (ops.empty()) return;
                op = std::move(ops.front());
                ops.pop_front();
            }
            if (max_delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % (unsigned)max_delay_us));
            op();
            {
                std::lock_guard<std::mutex> lk(mu);
                pending--;
            }
            cv.notify_all();
        }
    }
    void enqueue(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            ops.push_back(std::move(f));
            pending++;
        }
        cv.notify_all();
    }
    void sync() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return pending == 0; });
    }
};

// two-rank rendezvous shared by the ranks' devices
struct Fabric {
    std::mutex mu;
    std::condition_variable cv;
    const double *send[2] = {nullptr, nullptr};
    size_t count[2] = {0, 0};
    unsigned long arrived = 0, finished = 0;  // generation counters
    int nranks = 2;
};

struct FakeDev {
    typedef FakeStream *Stream;
    typedef FakeEvent *Event;
    int rank = 0;
    Fabric *fab = nullptr;
    std::atomic<int> *errors = nullptr;
    int record(Event e, Stream s) {
        unsigned long g;
        {
            std::lock_guard<std::mutex> lk(e->mu);
            g = ++e->recorded;
        }
        s->enqueue([e, g] {
            {
                std::lock_guard<std::mutex> lk(e->mu);
                if (e->done < g) e->done = g;
            }
            e->cv.notify_all();
        });
        return 0;
    }
    int wait(Stream s, Event e) {
        unsigned long g;
        {
            std::lock_guard<std::mutex> lk(e->mu);
            g = e->recorded;  // the most recent record at the time of the call
        }
        s->enqueue([e, g] {
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv.wait(lk, [e, g] { return e->done >= g; });
        });
        return 0;
    }
    int reduce(const double *send, double *recv, size_t count, int root, Stream s) {
        Fabric *f = fab;
        const int r = rank;
        std::atomic<int> *err = errors;
        s->enqueue([f, r, send, recv, count, root, err] {
            std::unique_lock<std::mutex> lk(f->mu);
            const unsigned long gen = f->arrived / (unsigned long)f->nranks;  // collectives are issued in the same order on every rank
            f->send[r] = send;
            f->count[r] = count;
            f->arrived++;
            f->cv.notify_all();
            f->cv.wait(lk, [f, gen] { return f->arrived >= (gen + 1) * (unsigned long)f->nranks; });
            if (r == root) {
                if (f->count[0] != f->count[1]) (*err)++;
                // read the send buffers slowly, so that a premature refill of a staging buffer is caught in the act
                lk.unlock();
                for (size_t i = 0; i < count; i++) {
                    recv[i] = f->send[0][i] + f->send[1][i];
                    if ((i & 63) == 0) std::this_thread::yield();
                }
                lk.lock();
                f->finished = gen + 1;
                f->cv.notify_all();
            } else {
                f->cv.wait(lk, [f, gen] { return f->finished >= gen + 1; });
            }
        });
        return 0;
    }
    int copy_to_host(double *h, const double *d, size_t count, Stream s) {
        s->enqueue([h, d, count] {
            for (size_t i = 0; i < count; i++) h[i] = d[i];
        });
        return 0;
    }
    int fold(const double *src, double *dst, size_t blocks, size_t groups, size_t block, Stream s) {
        s->enqueue([src, dst, blocks, groups, block] {
            for (size_t k = 0; k < blocks; k++)
                for (size_t i = 0; i < block; i++) {
                    double t = src[(k * groups) * block + i];
                    for (size_t g = 1; g < groups; g++) t += src[(k * groups + g) * block + i];
                    dst[k * block + i] = t;
                    if ((i & 63) == 0) std::this_thread::yield();  // (read slowly: a premature refill of the rows is caught)
                }
        });
        return 0;
    }
    bool is_root(int root) { return rank == root; }
};

double value(int rank, size_t k, size_t i) { return (double)(rank + 1) * 1000003.0 + (double)k * 17.0 + (double)i * 0.5; }

struct RankCtx {
    FakeDev dev;
    std::unique_ptr<FakeStream> caller, qs, reader;
    FakeEvent ev[6];
    mxg::MixQueueCore<FakeDev> q;
    std::vector<double> mem[4];
    std::vector<double> gmem[2];
    std::vector<double> sink;
};

// groups > 1: a slot is [groups][block] rows -- row 0 carries value(), row g > 0 the constant g (all exact), so the fold's sum is
// value + groups (groups - 1) / 2
int run_case(size_t block, int depth, size_t blocks, int delay_us, unsigned seed, bool verbose, size_t groups = 1) {
    Fabric fab;
    std::atomic<int> errors{0};
    RankCtx R[2];
    const size_t ring = (size_t)depth * 4;
    for (int r = 0; r < 2; r++) {
        RankCtx &c = R[r];
        c.dev.rank = r;
        c.dev.fab = &fab;
        c.dev.errors = &errors;
        c.caller.reset(new FakeStream(seed * 7 + r * 3 + 1, delay_us));
        c.qs.reset(new FakeStream(seed * 11 + r * 5 + 2, delay_us));
        c.reader.reset(new FakeStream(seed * 13 + r * 7 + 3, delay_us * 3));
        for (int b = 0; b < 4; b++) c.mem[b].assign(block * (size_t)depth, -1.0);
        for (int b = 0; b < 2; b++) c.gmem[b].assign(groups > 1 ? block * (size_t)depth * groups : 1, -5.0);
        c.q.groups = groups;
        c.q.dev = &c.dev;
        c.q.block = block;
        c.q.depth = depth;
        c.q.root = 0;
        for (int b = 0; b < 2; b++) {
            c.q.stage[b] = c.mem[b].data();
            c.q.result[b] = c.mem[2 + b].data();
            if (groups > 1) c.q.gstage[b] = c.gmem[b].data();
            c.q.filled[b] = &c.ev[b];
            c.q.reduced[b] = &c.ev[2 + b];
            c.q.consumed[b] = &c.ev[4 + b];
        }
        c.q.qstream = c.qs.get();
        if (r == 0) {
            c.sink.assign(ring * block, -2.0);
            c.q.h_sink = c.sink.data();
            c.q.sink_blocks = ring;
        }
    }
    // asynchronous reads of the root's results, ordered only by release()
    std::vector<double> seen(blocks * block, -3.0);
    std::vector<double> sink_copy(blocks * block, -4.0);
    // the host side of the two ranks: like two processes, each issuing slot / render / push without waiting for the device
    auto host = [&](int r) {
        RankCtx &c = R[r];
        size_t batch_first = 0;
        for (size_t k = 0; k < blocks; k++) {
            int st = 0;
            double *slot = c.q.slot(c.caller.get(), &st);
            if (st || !slot) {
                errors++;
                return;
            }
            c.caller->enqueue([slot, r, k, block, groups] {  // the "kernel" that renders + mixes block k into its slot
                for (size_t i = 0; i < block; i++) slot[i] = value(r, k, i);
                for (size_t g = 1; g < groups; g++)
                    for (size_t i = 0; i < block; i++) slot[g * block + i] = (double)g;
            });
            const size_t before = c.q.batches;
            if (c.q.push(c.caller.get())) errors++;
            if (c.q.batches != before && r == 0) {
                // a batch was submitted: read its sum asynchronously on a third stream, after its reduce (reduced[b]) and
                // declare the read with release() so that the reduce of the batch after next cannot overwrite it first
                const int b = c.q.last;
                const size_t nb = c.q.last_blocks;
                const double *res = c.q.result[b];
                c.dev.wait(c.reader.get(), c.q.reduced[b]);
                double *dst = seen.data() + batch_first * block;
                c.reader->enqueue([res, dst, nb, block] {
                    for (size_t i = 0; i < nb * block; i++) {
                        dst[i] = res[i];
                        if ((i & 31) == 0) std::this_thread::yield();
                    }
                });
                if (c.q.release(c.reader.get())) errors++;
                // the sink ring holds the last `ring` blocks: copy the batch out before the ring wraps (host-side consumer)
                if ((batch_first / (size_t)depth) % 2 == 1) {
                    c.qs->sync();
                    const size_t from = batch_first + nb >= ring ? batch_first + nb - ring : 0;
                    for (size_t kk = from; kk < batch_first + nb; kk++)
                        for (size_t i = 0; i < block; i++) sink_copy[kk * block + i] = c.sink[(kk % ring) * block + i];
                }
                batch_first += nb;
            } else if (c.q.batches != before) {
                batch_first += c.q.last_blocks;
            }
        }
        if (c.q.flush(c.caller.get())) errors++;
        if (r == 0 && c.q.fill == 0 && batch_first < blocks) {  // the partial last batch submitted by flush
            const int b = c.q.last;
            const size_t nb = c.q.last_blocks;
            c.dev.wait(c.reader.get(), c.q.reduced[b]);
            const double *res = c.q.result[b];
            double *dst = seen.data() + batch_first * block;
            c.reader->enqueue([res, dst, nb, block] {
                for (size_t i = 0; i < nb * block; i++) dst[i] = res[i];
            });
            c.q.release(c.reader.get());
        }
        c.caller->sync();
        c.qs->sync();
        c.reader->sync();
    };
    std::thread t1(host, 1);
    host(0);
    t1.join();
    // after flush + sync: result() of the last batch, every asynchronous read, the sink ring
    size_t bad_seen = 0, bad_sink = 0, bad_last = 0;
    const double gsum = (double)(groups * (groups - 1));  // both ranks' rows 1 .. groups - 1
    for (size_t k = 0; k < blocks; k++)
        for (size_t i = 0; i < block; i++) {
            const double want = value(0, k, i) + value(1, k, i) + gsum;
            if (seen[k * block + i] != want) bad_seen++;
            const double sc = sink_copy[k * block + i];
            if (sc != -4.0 && sc != want) bad_sink++;
        }
    {
        RankCtx &c = R[0];
        const size_t first = blocks - c.q.last_blocks;
        for (size_t k = 0; k < c.q.last_blocks; k++)
            for (size_t i = 0; i < block; i++)
                if (c.q.result[c.q.last][k * block + i] != value(0, first + k, i) + value(1, first + k, i) + gsum) bad_last++;
        // the ring's final content: the last `ring` blocks
        for (size_t k = blocks > ring ? blocks - ring : 0; k < blocks; k++)
            for (size_t i = 0; i < block; i++)
                if (c.sink[(k % ring) * block + i] != value(0, k, i) + value(1, k, i) + gsum) bad_sink++;
    }
    const size_t expect_batches = (blocks + (size_t)depth - 1) / (size_t)depth;
    int bad = errors.load() + (R[0].q.batches != expect_batches) + (R[1].q.batches != expect_batches);
    if (verbose || bad || bad_seen || bad_sink || bad_last)
        printf("case block=%zu depth=%d blocks=%zu groups=%zu delay=%dus seed=%u: batches %zu/%zu, wrong async reads %zu, wrong sink %zu, "
               "wrong last result %zu, protocol errors %d\n",
               block, depth, blocks, groups, delay_us, seed, R[0].q.batches, expect_batches, bad_seen, bad_sink, bad_last, errors.load());
    return bad + (bad_seen != 0) + (bad_sink != 0) + (bad_last != 0);
}

}  // namespace

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    int failed = 0, cases = 0;
    for (int rnd = 0; rnd < rounds; rnd++) {
        // (block doubles, depth M, blocks): full batches only, a partial last batch, depth 1 (config 5's one reduce per
        // render), long runs with many batches in flight
        const size_t cfg[][3] = {{64, 3, 7}, {1024, 16, 80}, {256, 1, 9}, {32, 4, 64}, {128, 5, 23}, {16, 2, 41}};
        for (auto &c : cfg) {
            failed += run_case(c[0], (int)c[1], c[2], rnd % 3 == 0 ? 0 : 40 * (rnd % 3), 1000u + (unsigned)rnd * 97u + (unsigned)cases,
                               false) != 0;
            cases++;
        }
        // grouped slots (the config-2 step: the queue folds [groups][block] rows per block on its own stream before the reduce)
        const size_t gcfg[][4] = {{64, 3, 7, 4}, {128, 16, 50, 3}, {32, 2, 21, 7}};
        for (auto &c : gcfg) {
            failed += run_case(c[0], (int)c[1], c[2], rnd % 3 == 0 ? 0 : 40 * (rnd % 3), 5000u + (unsigned)rnd * 89u + (unsigned)cases,
                               false, c[3]) != 0;
            cases++;
        }
    }
    printf("mixq protocol: %d cases, %d failed\n", cases, failed);
    return failed ? 1 : 0;
}
