// tests/mock_rccl/mock_rccl.cpp -- TEST INFRASTRUCTURE: a single-process stand-in for the ten RCCL entry points
// csrc/group.hip binds, so that the gather-v logic of mi355_group_gather (counts, rank-ordered places, matching of
// sends and receives, a root other than 0) can run with SEVERAL ranks on a box that has ONE GPU: real RCCL refuses
// two ranks on one device, this stand-in does not care.  It is loaded only when MI355_RCCL_LIB points at it
// (tests/test_group_gpu.py); the product never uses it.
//
// Semantics kept: operations issued between ncclGroupStart/End (or alone) take effect at the outermost
// ncclGroupEnd; an all-gather needs the call of every rank of the communicator in the same group; a send matches
// the receive posted by its peer for it, in order of issue per (sender, receiver); data moves on the receiver's
// stream after the sender's stream has drained.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct MockComm { int rank, nranks, device, world; };
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
};
std::vector<Op> g_ops;
int g_depth = 0, g_world = 0;

size_t elem(ncclDataType_t t) { return t == ncclInt8 || t == ncclUint8 ? 1 : t == ncclInt32 || t == ncclUint32 ? 4 : 8; }

ncclResult_t move(const void *src, hipStream_t sstream, int sdev, void *dst, hipStream_t dstream, int ddev, size_t bytes) {
    if (hipSetDevice(sdev) != hipSuccess || hipStreamSynchronize(sstream) != hipSuccess) return ncclInternalError;
    if (hipSetDevice(ddev) != hipSuccess) return ncclInternalError;
    if (bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dstream) != hipSuccess) return ncclInternalError;
    return ncclSuccess;
}

ncclResult_t flush() {
    std::vector<Op> ops;
    ops.swap(g_ops);
    // all-gathers: every rank of the world must have called
    std::map<int, std::vector<Op *>> ag;
    for (Op &o : ops) if (o.kind == 0) ag[o.comm->world].push_back(&o);
    for (auto &kv : ag) {
        std::vector<Op *> &v = kv.second;
        if ((int)v.size() != v[0]->comm->nranks) { fprintf(stderr, "mock rccl: all-gather with %zu of %d ranks\n", v.size(), v[0]->comm->nranks); return ncclInvalidArgument; }
        for (Op *r : v)
            for (Op *q : v)
                if (move(q->src, q->stream, q->comm->device, (char *)r->dst + (size_t)q->comm->rank * q->bytes, r->stream, r->comm->device, q->bytes) != ncclSuccess) return ncclInternalError;
    }
    // point to point: FIFO per (world, sender, receiver)
    std::map<std::vector<int>, std::deque<Op *>> sends;
    for (Op &o : ops) if (o.kind == 1) sends[{o.comm->world, o.comm->rank, o.peer}].push_back(&o);
    for (Op &o : ops) {
        if (o.kind != 2) continue;
        auto &q = sends[{o.comm->world, o.peer, o.comm->rank}];
        if (q.empty()) { fprintf(stderr, "mock rccl: receive at rank %d from %d has no send\n", o.comm->rank, o.peer); return ncclInvalidArgument; }
        Op *s = q.front();
        q.pop_front();
        if (s->bytes != o.bytes) { fprintf(stderr, "mock rccl: send of %zu bytes meets receive of %zu\n", s->bytes, o.bytes); return ncclInvalidArgument; }
        if (move(s->src, s->stream, s->comm->device, o.dst, o.stream, o.comm->device, o.bytes) != ncclSuccess) return ncclInternalError;
    }
    for (auto &kv : sends)
        if (!kv.second.empty()) { fprintf(stderr, "mock rccl: %zu unmatched sends\n", kv.second.size()); return ncclInvalidArgument; }
    return ncclSuccess;
}

ncclResult_t issue(const Op &o) {
    g_ops.push_back(o);
    return g_depth ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5a, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    const int world = ++g_world;
    for (int i = 0; i < ndev; i++) comms[i] = new MockComm{i, ndev, devlist ? devlist[i] : i, world};
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank) {
    if (nranks != 1) { fprintf(stderr, "mock rccl: one process only\n"); return ncclInvalidArgument; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    *comm = new MockComm{rank, nranks, dev, ++g_world};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return --g_depth == 0 ? flush() : ncclSuccess; }
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
