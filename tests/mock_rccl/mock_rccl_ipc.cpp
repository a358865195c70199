// tests/mock_rccl/mock_rccl_ipc.cpp -- TEST INFRASTRUCTURE: the inter-PROCESS stand-in for the ten RCCL entry points
// csrc/group.hip binds.  mock_rccl.cpp lets several ranks of ONE process share the one GPU of a test box; this file
// does the same for one rank per PROCESS -- the shape `python -m torch.distributed.run ... bench.py --gpus N` has on the
// 8-GPU node (every process: its own HIP context, its own copy of the library, mi355_group_adopt_rank with an id handed
// around by the launcher) -- so that exactly that sequence can run on a one-GPU box before the first real multi-GPU
// run.  Real RCCL refuses two ranks on one device; this stand-in does not care.  It is loaded only when MI355_RCCL_LIB
// points at it, and needs MOCK_RCCL_SHM=/name (a POSIX shared-memory object all ranks of the job agree on; the test that
// starts the ranks unlinks it).  The product never uses it.
//
// How the ranks meet: a shared-memory segment holds a process-shared mutex / condition variable, a table of
// communicators ("worlds", keyed by the 128-byte id) and a table of messages; payloads are staged through an arena in
// the same segment (device -> shared host memory by the sender, shared host memory -> device by the receiver: slow and
// simple -- HIP IPC handles would need the dmabuf export path of the pool's driver and prove nothing about group.hip).
// Semantics kept from RCCL: operations issued between ncclGroupStart/End (or alone) take effect at the outermost
// ncclGroupEnd; the k-th all-gather of a rank meets the k-th all-gather of every other rank of the communicator; a send
// matches the receive its peer posts for it, in order of issue per (sender, receiver); data lands on the receiver's
// stream after the sender's stream has drained; ncclGroupEnd returns when the rank's own operations have completed.  A
// rank whose partner never comes gets an error after MOCK_RCCL_TIMEOUT_S seconds (default 20) instead of hanging the box.
#include <hip/hip_runtime.h>

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct MockComm { int rank, nranks, device, world; };
typedef MockComm *ncclComm_t;
}

namespace {

constexpr int kMaxWorlds = 16, kMaxRanks = 16, kMaxMsgs = 1024;
constexpr uint32_t kMagic = 0x600DF00Du;

struct Msg {
    uint32_t state;   // 0 free, 1 posted (payload staged), 3 reserved (its sender is staging the payload)
    uint32_t world, src, dst, channel;   // channel 0: all-gather pieces, 1: point to point
    uint64_t seq, bytes, arena_off;
    uint64_t uid;   // never reused: a sender that waits for ITS message to be taken must not mistake the slot's next tenant for it
};

struct Shm {
    std::atomic<uint32_t> ready;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    uint32_t id_counter;
    uint64_t uid_counter;
    struct World { char key[128]; uint32_t used, nranks, joined; } worlds[kMaxWorlds];
    Msg msgs[kMaxMsgs];
    uint64_t seq_post[kMaxWorlds][kMaxRanks][kMaxRanks][2];   // next sequence number a sender gives out
    uint64_t seq_take[kMaxWorlds][kMaxRanks][kMaxRanks][2];   // next sequence number the receiver takes
    uint64_t arena_bytes;   // payloads live in [arena_off, arena_off + bytes) of their message slot: first fit over the live slots
};

Shm *g_shm = nullptr;
uint8_t *g_arena = nullptr;

int timeout_s() {
    const char *e = getenv("MOCK_RCCL_TIMEOUT_S");
    return e && atoi(e) > 0 ? atoi(e) : 20;
}

bool attach() {
    if (g_shm) return true;
    const char *name = getenv("MOCK_RCCL_SHM");
    if (!name || name[0] != '/') {
        fprintf(stderr, "mock rccl (ipc): MOCK_RCCL_SHM=/name is not set\n");
        return false;
    }
    const char *mb = getenv("MOCK_RCCL_SHM_MB");
    const size_t arena = (size_t)(mb && atoi(mb) > 0 ? atoi(mb) : 64) << 20;
    const size_t head = (sizeof(Shm) + 4095) & ~(size_t)4095, total = head + arena;
    bool creator = true;
    int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 && errno == EEXIST) { creator = false; fd = shm_open(name, O_RDWR, 0600); }
    if (fd < 0) { perror("mock rccl (ipc): shm_open"); return false; }
    if (creator && ftruncate(fd, (off_t)total) != 0) { perror("mock rccl (ipc): ftruncate"); close(fd); return false; }
    for (int i = 0; !creator && i < 2000; i++) {   // the creator may not have sized the object yet
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= total) break;
        usleep(5000);
    }
    void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { perror("mock rccl (ipc): mmap"); return false; }
    Shm *s = (Shm *)p;
    if (creator) {
        pthread_mutexattr_t ma;
        pthread_mutexattr_init(&ma);
        pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
        pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
        pthread_mutex_init(&s->mu, &ma);
        pthread_condattr_t ca;
        pthread_condattr_init(&ca);
        pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
        pthread_condattr_setclock(&ca, CLOCK_MONOTONIC);
        pthread_cond_init(&s->cv, &ca);
        s->arena_bytes = arena;
        s->ready.store(kMagic, std::memory_order_release);
    } else {
        for (int i = 0; i < 4000 && s->ready.load(std::memory_order_acquire) != kMagic; i++) usleep(5000);
        if (s->ready.load(std::memory_order_acquire) != kMagic) { fprintf(stderr, "mock rccl (ipc): segment never became ready\n"); return false; }
    }
    g_shm = s;
    g_arena = (uint8_t *)p + head;
    return true;
}

struct Lock {
    Lock() { if (pthread_mutex_lock(&g_shm->mu) == EOWNERDEAD) pthread_mutex_consistent(&g_shm->mu); }
    ~Lock() { pthread_mutex_unlock(&g_shm->mu); }
    // waits until pred() or the deadline; with the lock held on entry and on return
    template <class P>
    bool wait(P pred) {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        ts.tv_sec += timeout_s();
        while (!pred()) {
            const int rc = pthread_cond_timedwait(&g_shm->cv, &g_shm->mu, &ts);
            if (rc == EOWNERDEAD) pthread_mutex_consistent(&g_shm->mu);
            if (rc == ETIMEDOUT) return pred();
        }
        return true;
    }
};

struct Op {
    int kind;   // 0 all-gather, 1 send, 2 recv
    MockComm *comm;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    hipStream_t stream;
};
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;

size_t elem(ncclDataType_t t) { return t == ncclInt8 || t == ncclUint8 ? 1 : t == ncclInt32 || t == ncclUint32 ? 4 : 8; }

// With the lock held: a free message slot and a gap of `need` bytes in the arena (first fit between the live payloads).
bool reserve(size_t need, int &slot, uint64_t &off) {
    slot = -1;
    struct Span { uint64_t off, len; } live[kMaxMsgs];
    int n = 0;
    for (int i = 0; i < kMaxMsgs; i++) {
        const Msg &m = g_shm->msgs[i];
        if (!m.state) { if (slot < 0) slot = i; continue; }
        live[n++] = Span{m.arena_off, (m.bytes + 255) & ~(uint64_t)255};
    }
    if (slot < 0) return false;
    for (int i = 1; i < n; i++)   // insertion sort by offset (a few dozen at most)
        for (int j = i; j > 0 && live[j].off < live[j - 1].off; j--) { const Span t = live[j]; live[j] = live[j - 1]; live[j - 1] = t; }
    uint64_t at = 0;
    for (int i = 0; i < n; i++) {
        if (live[i].off >= at + need) break;
        at = live[i].off + live[i].len > at ? live[i].off + live[i].len : at;
    }
    if (at + need > g_shm->arena_bytes) return false;
    off = at;
    return true;
}

// Stages `bytes` of device memory for (world, me -> dst, channel); returns the message slot or -1.
int post(MockComm *c, int dst, int channel, const void *src, size_t bytes, hipStream_t stream, uint64_t *uid = nullptr) {
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return -1;
    uint64_t off = 0;
    int slot = -1;
    {
        Lock lk;
        const size_t need = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
        if (need > g_shm->arena_bytes) { fprintf(stderr, "mock rccl (ipc): a message of %zu bytes exceeds the arena (MOCK_RCCL_SHM_MB)\n", bytes); return -1; }
        if (!lk.wait([&] { return reserve(need, slot, off); })) {
            fprintf(stderr, "mock rccl (ipc): rank %d: no room in the arena or the message table (timed out)\n", c->rank);
            return -1;
        }
        g_shm->msgs[slot] = Msg{3u, (uint32_t)c->world, (uint32_t)c->rank, (uint32_t)dst, (uint32_t)channel, 0, bytes, off, ++g_shm->uid_counter};
    }
    const bool ok = !bytes || hipMemcpy(g_arena + off, src, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    Lock lk;
    Msg &m = g_shm->msgs[slot];
    if (!ok) { m.state = 0; pthread_cond_broadcast(&g_shm->cv); return -1; }
    m.seq = g_shm->seq_post[c->world][c->rank][dst][channel]++;
    m.state = 1;
    if (uid) *uid = m.uid;
    pthread_cond_broadcast(&g_shm->cv);
    return slot;
}

void release(Msg &m) {   // with the lock held: the payload leaves the arena
    m.state = 0;
    pthread_cond_broadcast(&g_shm->cv);
}

// With the lock held: the slot of the next message of (world, src -> me, channel), or -1.
int find_msg(MockComm *c, int src, int channel) {
    const uint64_t want = g_shm->seq_take[c->world][src][c->rank][channel];
    for (int i = 0; i < kMaxMsgs; i++) {
        const Msg &m = g_shm->msgs[i];
        if (m.state == 1 && (int)m.world == c->world && (int)m.src == src && (int)m.dst == c->rank && (int)m.channel == channel && m.seq == want) return i;
    }
    return -1;
}

// Moves message `slot` into device memory on `stream` and frees it.
bool consume(MockComm *c, int slot, int src, int channel, void *dst, size_t bytes, hipStream_t stream) {
    Msg &m = g_shm->msgs[slot];
    bool ok = m.bytes == bytes;
    if (!ok) fprintf(stderr, "mock rccl (ipc): send of %llu bytes meets receive of %zu\n", (unsigned long long)m.bytes, bytes);
    ok = ok && hipSetDevice(c->device) == hipSuccess;
    if (ok && bytes) ok = hipMemcpyAsync(dst, g_arena + m.arena_off, bytes, hipMemcpyHostToDevice, stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(stream) == hipSuccess;   // the arena chunk is reused once released
    Lock lk;
    g_shm->seq_take[c->world][src][c->rank][channel]++;
    release(m);
    return ok;
}

// What a rank still has to receive; taken in whatever order the messages arrive (the senders share one arena: waiting
// for rank 1's payload while the arena is full of rank 2's and 3's would never end).
struct Want { MockComm *c; int src, channel; void *dst; size_t bytes; hipStream_t stream; bool done; };

bool take_all(std::vector<Want> &wants) {
    for (size_t left = wants.size(); left; left--) {
        int slot = -1;
        Want *w = nullptr;
        {
            Lock lk;
            auto any = [&] {
                for (Want &x : wants)
                    if (!x.done && (slot = find_msg(x.c, x.src, x.channel)) >= 0) { w = &x; return true; }
                return false;
            };
            if (!lk.wait(any)) {
                for (Want &x : wants)
                    if (!x.done) fprintf(stderr, "mock rccl (ipc): rank %d: %s from rank %d never met its partner (timed out)\n", x.c->rank,
                                         x.channel ? "receive" : "all-gather piece", x.src);
                return false;
            }
        }
        w->done = true;
        if (!consume(w->c, slot, w->src, w->channel, w->dst, w->bytes, w->stream)) return false;
    }
    return true;
}

ncclResult_t flush() {
    std::vector<Op> ops;
    ops.swap(t_ops);
    if (ops.empty()) return ncclSuccess;
    bool ok = true;
    struct Sent { int slot; uint64_t uid; };
    std::vector<Sent> posted;
    // 1. everything this rank sends (an all-gather: its piece to every other rank)
    for (Op &o : ops) {
        uint64_t uid = 0;
        if (o.kind == 1) { const int s = post(o.comm, o.peer, 1, o.src, o.bytes, o.stream, &uid); ok &= s >= 0; if (s >= 0) posted.push_back(Sent{s, uid}); }
        if (o.kind == 0)
            for (int r = 0; r < o.comm->nranks && ok; r++) {
                if (r == o.comm->rank) continue;
                const int s = post(o.comm, r, 0, o.src, o.bytes, o.stream, &uid);
                ok &= s >= 0;
                if (s >= 0) posted.push_back(Sent{s, uid});
            }
    }
    // 2. everything it receives.  Several receives from ONE sender on one channel complete in the order of issue (each
    //    waits for that sender's next sequence number); across senders, in the order of arrival.
    std::vector<Want> wants;
    for (Op &o : ops) {
        if (o.kind == 2) wants.push_back(Want{o.comm, o.peer, 1, o.dst, o.bytes, o.stream, false});
        if (o.kind == 0) {
            ok = ok && hipSetDevice(o.comm->device) == hipSuccess &&
                 hipMemcpyAsync((char *)o.dst + (size_t)o.comm->rank * o.bytes, o.src, o.bytes, hipMemcpyDeviceToDevice, o.stream) == hipSuccess;
            for (int r = 0; r < o.comm->nranks; r++)
                if (r != o.comm->rank) wants.push_back(Want{o.comm, r, 0, (char *)o.dst + (size_t)r * o.bytes, o.bytes, o.stream, false});
        }
    }
    // (receives of one (sender, channel) must be consumed in the order they were posted: find_msg hands out the next
    // sequence number, and take_all gives it to the FIRST unfinished want of that sender -- which is the oldest)
    if (ok) ok = take_all(wants);
    // 3. its sends must have been taken (RCCL's group end returns when the rank's operations are complete); what nobody
    //    took is withdrawn
    {
        Lock lk;
        auto mine = [&](const Sent &p) { const Msg &m = g_shm->msgs[p.slot]; return m.state == 1 && m.uid == p.uid; };
        auto all_taken = [&] { for (const Sent &p : posted) if (mine(p)) return false; return true; };
        if (!lk.wait(all_taken)) {
            for (const Sent &p : posted) {
                Msg &m = g_shm->msgs[p.slot];
                if (!mine(p)) continue;
                fprintf(stderr, "mock rccl (ipc): rank %u: %s to rank %u never met its partner (timed out)\n", m.src, m.channel ? "send" : "all-gather piece", m.dst);
                g_shm->seq_take[m.world][m.src][m.dst][m.channel]++;   // the receiver will not see this one any more
                release(m);
            }
            ok = false;
        }
    }
    return ok ? ncclSuccess : ncclInternalError;
}

ncclResult_t issue(const Op &o) {
    if (!attach()) return ncclInternalError;
    t_ops.push_back(o);
    return t_depth ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!attach()) return ncclInternalError;
    Lock lk;
    memset(id, 0x5a, sizeof *id);
    const uint32_t c = ++g_shm->id_counter, pid = (uint32_t)getpid();
    memcpy(id->internal, &c, sizeof c);
    memcpy(id->internal + 4, &pid, sizeof pid);
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) {
    fprintf(stderr, "mock rccl (ipc): ncclCommInitAll is the one-process form: use tests/mock_rccl/librccl_mock.so\n");
    return ncclInvalidArgument;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!attach()) return ncclInternalError;
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Lock lk;
    int w = -1;
    for (int i = 0; i < kMaxWorlds; i++)
        if (g_shm->worlds[i].used && !memcmp(g_shm->worlds[i].key, id.internal, sizeof id.internal)) w = i;
    for (int i = 0; w < 0 && i < kMaxWorlds; i++)
        if (!g_shm->worlds[i].used) {
            w = i;
            g_shm->worlds[i].used = 1;
            g_shm->worlds[i].nranks = (uint32_t)nranks;
            g_shm->worlds[i].joined = 0;
            memcpy(g_shm->worlds[i].key, id.internal, sizeof id.internal);
        }
    if (w < 0) { fprintf(stderr, "mock rccl (ipc): communicator table full\n"); return ncclInternalError; }
    if ((int)g_shm->worlds[w].nranks != nranks) { fprintf(stderr, "mock rccl (ipc): rank %d: the ranks disagree about nranks\n", rank); return ncclInvalidArgument; }
    g_shm->worlds[w].joined++;
    pthread_cond_broadcast(&g_shm->cv);
    // like RCCL: communicator creation is collective
    if (!lk.wait([&] { return (int)g_shm->worlds[w].joined >= nranks; })) {
        fprintf(stderr, "mock rccl (ipc): rank %d: only %u of %d ranks joined the communicator (timed out)\n", rank, g_shm->worlds[w].joined, nranks);
        return ncclInternalError;
    }
    *comm = new MockComm{rank, nranks, dev, w};
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
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl (ipc) error (see stderr)"; }

}  // extern "C"
