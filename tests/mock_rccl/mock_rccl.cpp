// tests/mock_rccl/mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the ten RCCL entry points csrc/group.hip
// binds, so that the gather-v logic of mi355_group_gather (counts, rank-ordered places, matching of sends and
// receives, a root other than 0, the collective capacity verdict) can run with SEVERAL ranks on a box that has ONE
// GPU: real RCCL refuses two ranks on one device, this stand-in does not care.  It is loaded only when
// MI355_RCCL_LIB points at it (tests/test_group_gpu.py); the product never uses it.
//
// Two shapes of a communicator, as in RCCL:
//   * ncclCommInitAll   -- every rank in ONE thread (mi355_group_create): a group's operations are matched when the
//                          outermost ncclGroupEnd runs; anything left unmatched is an error at once;
//   * ncclCommInitRank  -- one rank per THREAD (mi355_group_adopt_rank, the one-member-per-process form of
//                          bench.py under torch.distributed.run, with threads standing in for the processes): a
//                          thread's outermost ncclGroupEnd posts its operations and BLOCKS until its peers have
//                          posted the matching ones -- a rank whose peers never call is reported after
//                          MOCK_RCCL_TIMEOUT_S seconds (default 20) as an error instead of hanging the test box.
//
// Semantics kept: operations issued between ncclGroupStart/End (or alone) take effect at the outermost
// ncclGroupEnd; the k-th all-gather of a rank meets the k-th all-gather of every other rank of the communicator;
// a send matches the receive posted by its peer for it, in order of issue per (sender, receiver); data moves on
// the receiver's stream after the sender's stream has drained.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct MockComm { int rank, nranks, device, world; bool threaded; };
typedef MockComm *ncclComm_t;

}  // extern "C"

namespace {

struct Op {
    int kind;   // 0 all-gather, 1 send, 2 recv
    MockComm *comm;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    hipStream_t stream;
    bool done = false, failed = false;
};
typedef std::shared_ptr<Op> OpP;

std::mutex g_mu;
std::condition_variable g_cv;
int g_next_world = 0;
std::map<std::string, int> g_world_of_id;
// pending operations, per world: all-gathers FIFO per rank, sends FIFO per (sender, receiver), receives likewise
struct World {
    std::map<int, std::deque<OpP>> ag;
    std::map<std::pair<int, int>, std::deque<OpP>> sends, recvs;
};
std::map<int, World> g_worlds;

thread_local std::vector<OpP> t_ops;
thread_local int t_depth = 0;

size_t elem(ncclDataType_t t) { return t == ncclInt8 || t == ncclUint8 ? 1 : t == ncclInt32 || t == ncclUint32 ? 4 : 8; }

bool move(const void *src, hipStream_t sstream, int sdev, void *dst, hipStream_t dstream, int ddev, size_t bytes) {
    if (hipSetDevice(sdev) != hipSuccess || hipStreamSynchronize(sstream) != hipSuccess) return false;
    if (hipSetDevice(ddev) != hipSuccess) return false;
    if (bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dstream) != hipSuccess) return false;
    // the receiver may run on another thread: its stream must hold the copy before that thread goes on
    return true;
}

// Called with g_mu held: completes everything that has found its partners.
void match_all() {
    for (auto &wkv : g_worlds) {
        World &w = wkv.second;
        // all-gathers: the front of every rank's queue
        for (;;) {
            if (w.ag.empty()) break;
            const int nranks = w.ag.begin()->second.empty() ? 0 : w.ag.begin()->second.front()->comm->nranks;
            bool ready = nranks > 0 && (int)w.ag.size() == nranks;
            if (ready) for (auto &kv : w.ag) ready &= !kv.second.empty();
            if (!ready) break;
            std::vector<OpP> v;
            for (auto &kv : w.ag) { v.push_back(kv.second.front()); kv.second.pop_front(); }
            bool ok = true;
            for (OpP &r : v)
                for (OpP &q : v) {
                    ok &= q->bytes == r->bytes;
                    ok = ok && move(q->src, q->stream, q->comm->device, (char *)r->dst + (size_t)q->comm->rank * q->bytes, r->stream,
                                    r->comm->device, q->bytes);
                }
            for (OpP &r : v) { r->done = true; r->failed = !ok; }
            if (!ok) fprintf(stderr, "mock rccl: all-gather failed (sizes differ or a copy failed)\n");
        }
        // point to point: FIFO per (sender, receiver)
        for (auto &kv : w.recvs) {
            auto &rq = kv.second;
            auto &sq = w.sends[{kv.first.first, kv.first.second}];
            while (!rq.empty() && !sq.empty()) {
                OpP r = rq.front(), s = sq.front();
                rq.pop_front();
                sq.pop_front();
                bool ok = s->bytes == r->bytes;
                if (!ok) fprintf(stderr, "mock rccl: send of %zu bytes meets receive of %zu\n", s->bytes, r->bytes);
                ok = ok && move(s->src, s->stream, s->comm->device, r->dst, r->stream, r->comm->device, r->bytes);
                r->done = s->done = true;
                r->failed = s->failed = !ok;
            }
        }
    }
}

void withdraw(const OpP &o) {   // with g_mu held: an operation that will never complete leaves the pool
    World &w = g_worlds[o->comm->world];
    auto drop = [&](std::deque<OpP> &q) { for (auto it = q.begin(); it != q.end(); ++it) if (*it == o) { q.erase(it); break; } };
    if (o->kind == 0) drop(w.ag[o->comm->rank]);
    else if (o->kind == 1) drop(w.sends[{o->comm->rank, o->peer}]);
    else drop(w.recvs[{o->peer, o->comm->rank}]);
}

ncclResult_t flush() {
    std::vector<OpP> ops;
    ops.swap(t_ops);
    if (ops.empty()) return ncclSuccess;
    std::unique_lock<std::mutex> lk(g_mu);
    bool threaded = false;
    for (OpP &o : ops) {
        World &w = g_worlds[o->comm->world];
        threaded |= o->comm->threaded;
        if (o->kind == 0) w.ag[o->comm->rank].push_back(o);
        else if (o->kind == 1) w.sends[{o->comm->rank, o->peer}].push_back(o);
        else w.recvs[{o->peer, o->comm->rank}].push_back(o);
    }
    match_all();
    g_cv.notify_all();
    auto all_done = [&]() { for (OpP &o : ops) if (!o->done) return false; return true; };
    if (!all_done()) {
        if (threaded) {
            const char *e = getenv("MOCK_RCCL_TIMEOUT_S");
            const int secs = e && atoi(e) > 0 ? atoi(e) : 20;
            g_cv.wait_for(lk, std::chrono::seconds(secs), all_done);
        }
        if (!all_done()) {
            for (OpP &o : ops)
                if (!o->done) {
                    fprintf(stderr, "mock rccl: rank %d: %s %s rank %d never met its partner%s\n", o->comm->rank,
                            o->kind == 0 ? "all-gather" : o->kind == 1 ? "send" : "receive", o->kind == 1 ? "to" : "from", o->peer,
                            threaded ? " (timed out)" : "");
                    withdraw(o);
                }
            return ncclInvalidArgument;
        }
    }
    for (OpP &o : ops) if (o->failed) return ncclInternalError;
    // the copies were enqueued on the receivers' streams, possibly by another thread: make them visible to a
    // host that synchronises its own stream next (same device, in-order streams: nothing more to do)
    return ncclSuccess;
}

ncclResult_t issue(Op o) {
    t_ops.push_back(std::make_shared<Op>(o));
    return t_depth ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    static int counter = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0x5a, sizeof *id);
    const int c = ++counter;
    memcpy(id->internal, &c, sizeof c);
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int world = ++g_next_world;
    for (int i = 0; i < ndev; i++) comms[i] = new MockComm{i, ndev, devlist ? devlist[i] : i, world, false};
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_mu);
    const std::string key(id.internal, sizeof id.internal);
    auto it = g_world_of_id.find(key);
    const int world = it != g_world_of_id.end() ? it->second : (g_world_of_id[key] = ++g_next_world);
    *comm = new MockComm{rank, nranks, dev, world, nranks > 1};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return --t_depth == 0 ? flush() : ncclSuccess; }
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    return issue(Op{0, comm, send, recv, count * elem(t), -1, s});
}
ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return issue(Op{1, comm, send, nullptr, count * elem(t), peer, s});
}
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return issue(Op{2, comm, nullptr, recv, count * elem(t), peer, s});
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error (see stderr)"; }

}  // extern "C"
