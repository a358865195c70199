// group.hip -- several GPUs of one node behind the C-ABI (include/mi355diff.h, "multi-GPU"): one core per device,
// RCCL over xGMI only for the final changed-pixel gather.
//
// The reference has no multi-GPU code (server/src/kernels.cu:385 uses device 0); this is BASELINE.json's
// north star for the path: "independent frames shard trivially across the 8 GPUs of one node with RCCL over
// xGMI only for the final changed-pixel gather".  The data path has no collective: every member runs its own
// stream, its own dealt-out frame pairs (E1 round robin, BASELINE config 5) or its row band
// (E2) through the ordinary single-device entry points, concurrently, each on its own device and stream.
//
// The gather is a gather-v of (per-frame index, xs, diff) to one root:
//   1. ncclAllGather of four words per rank {entries of its batch, entries its buffers hold, the root's capacity,
//      argument check}: everybody learns every total AND reaches the same verdict on the capacities before
//      anything is sent -- a gather that cannot work fails on every rank, nobody is left sending to a root that
//      has already returned (one host synchronisation, as the reference reads h_pos back before its copies,
//      kernels.cu:507-508);
//   2. inside one ncclGroupStart/End: every other rank ncclSend's its index, xs and diff to the root, the root
//      posts the matching ncclRecv's at rank-ordered places -- up to 7 concurrent point-to-point transfers into
//      the root over its 7 direct xGMI links, no ring;
//   3. the root's own part is a device-to-device copy.
//
// A group lives either in one process over several devices (mi355_group_create: ncclCommInitAll; a C++ server
// linked against libmi355compat.a) or as one member per process (mi355_group_adopt_rank: ncclCommInitRank with an
// id made by mi355_group_unique_id and handed around by the launcher; bench.py under torch.distributed.run).
//
// RCCL is bound at the first group call with dlopen("librccl.so.1"): a copy the process has already loaded
// (PyTorch ships one) is reused, so that there is one RCCL per process; single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mi355diff.h"
#include "internal.h"

using namespace mi355;

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;

int bind_rccl() {
    if (g_rccl.handle) return MI355_OK;
    // MI355_RCCL_LIB: another library with the same ten entry points (the tests' single-process stand-in, which
    // lets several ranks share the one GPU of a test box; tests/mock_rccl)
    const char *other = getenv("MI355_RCCL_LIB");
    void *h = other && *other ? dlopen(other, RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (other && *other && !h) return set_error(MI355_ERR_STATE, "MI355_RCCL_LIB is set but cannot be loaded");
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);   // the copy the process already has, if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return set_error(MI355_ERR_STATE, "librccl.so.1 not found: the multi-GPU entry points need RCCL");
    Rccl r;
    r.handle = h;
#define BIND(name)                                                                                     \
    r.name = (decltype(r.name))dlsym(h, "nccl" #name);                                                 \
    if (!r.name) return set_error(MI355_ERR_STATE, "librccl.so.1 lacks nccl" #name);
    BIND(GetUniqueId) BIND(CommInitRank) BIND(CommInitAll) BIND(CommDestroy) BIND(AllGather) BIND(Send) BIND(Recv)
    BIND(GroupStart) BIND(GroupEnd) BIND(GetErrorString)
#undef BIND
    g_rccl = r;
    return MI355_OK;
}

int rccl_fail(const char *what, ncclResult_t e) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error");
    return set_error(MI355_ERR_HIP, buf);
}

#define RCCL_TRY(expr)                                           \
    do {                                                         \
        ncclResult_t _e = (expr);                                \
        if (_e != ncclSuccess) return rccl_fail(#expr, _e);      \
    } while (0)
#define HIP_TRY_G(expr)                                                    \
    do {                                                                   \
        hipError_t _e = (expr);                                            \
        if (_e != hipSuccess) {                                            \
            char _b[256];                                                  \
            snprintf(_b, sizeof _b, "%s: %s", #expr, hipGetErrorString(_e)); \
            return set_error(MI355_ERR_HIP, _b);                           \
        }                                                                  \
    } while (0)

struct Member {
    mi355_core *core = nullptr;
    bool owned = false;
    int rank = 0;              // rank in the group
    ncclComm_t comm = nullptr;
    uint32_t *d_counts = nullptr;   // [nranks][kWords] on the member's device: what every rank published for the gather
    uint32_t *d_send = nullptr;     // [kWords] this member's own words
    uint32_t h_send[4] = {0, 0, 0, 0};
};
constexpr int kWords = 4;   // per rank: entries of its batch, its member capacity, the root's capacity, argument check

}  // namespace

struct mi355_group {
    int nranks = 0;
    std::vector<Member> local;
    std::vector<uint32_t> h_counts;
    std::vector<uint32_t> h_all;    // [nranks][kWords]
};

namespace {

Member *member_of_rank(mi355_group *g, int rank) {
    for (Member &m : g->local)
        if (m.rank == rank) return &m;
    return nullptr;
}

int check_group(const mi355_group *g) {
    if (!g) return set_error(MI355_ERR_INVALID, "null group");
    return MI355_OK;
}

}  // namespace

extern "C" {

int mi355_group_unique_id(void *id128) {
    if (!id128) return set_error(MI355_ERR_INVALID, "null argument");
    if (int rc = bind_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == MI355_GROUP_ID_BYTES, "id size");
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return MI355_OK;
}

void mi355_group_destroy(mi355_group *g) {
    if (!g) return;
    for (Member &m : g->local) {
        if (m.core) {
            (void)hipSetDevice(core_device(m.core));
            (void)hipStreamSynchronize(core_stream(m.core));
        }
        if (m.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m.comm);
        if (m.d_counts) (void)hipFree(m.d_counts);
        if (m.d_send) (void)hipFree(m.d_send);
        if (m.owned) mi355_destroy(m.core);
    }
    delete g;
}

int mi355_group_create(const mi355_config *cfg, int ndev, const int *devices, mi355_group **out) {
    if (!cfg || !out) return set_error(MI355_ERR_INVALID, "null argument");
    *out = nullptr;
    if (ndev < 1 || ndev > 64) return set_error(MI355_ERR_INVALID, "ndev outside [1, 64]");
    if (int rc = bind_rccl()) return rc;
    mi355_group *g = new (std::nothrow) mi355_group;
    if (!g) return set_error(MI355_ERR_INVALID, "out of host memory");
    g->nranks = ndev;
    g->local.resize(ndev);
    g->h_counts.resize(ndev);
    g->h_all.resize((size_t)ndev * kWords);
    std::vector<int> devs(ndev);
    int rc = MI355_OK;
    for (int i = 0; i < ndev && !rc; i++) {
        devs[i] = devices ? devices[i] : i;
        mi355_config c = *cfg;
        c.device = devs[i];
        Member &m = g->local[i];
        m.rank = i;
        m.owned = true;
        rc = mi355_create(&c, &m.core);
        if (!rc && hipSetDevice(devs[i]) != hipSuccess) rc = set_error(MI355_ERR_HIP, "hipSetDevice");
        if (!rc && hipMalloc((void **)&m.d_counts, sizeof(uint32_t) * kWords * ndev) != hipSuccess) rc = set_error(MI355_ERR_HIP, "hipMalloc");
        if (!rc && hipMalloc((void **)&m.d_send, sizeof(uint32_t) * kWords) != hipSuccess) rc = set_error(MI355_ERR_HIP, "hipMalloc");
    }
    if (!rc) {
        std::vector<ncclComm_t> comms(ndev);
        const ncclResult_t e = g_rccl.CommInitAll(comms.data(), ndev, devs.data());
        if (e != ncclSuccess) rc = rccl_fail("ncclCommInitAll", e);
        else for (int i = 0; i < ndev; i++) g->local[i].comm = comms[i];
    }
    if (rc) { mi355_group_destroy(g); return rc; }
    *out = g;
    return MI355_OK;
}

int mi355_group_adopt_rank(mi355_core *core, int nranks, int rank, const void *id128, mi355_group **out) {
    if (!core || !id128 || !out) return set_error(MI355_ERR_INVALID, "null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return set_error(MI355_ERR_INVALID, "rank outside [0, nranks)");
    if (int rc = bind_rccl()) return rc;
    mi355_group *g = new (std::nothrow) mi355_group;
    if (!g) return set_error(MI355_ERR_INVALID, "out of host memory");
    g->nranks = nranks;
    g->local.resize(1);
    g->h_counts.resize(nranks);
    g->h_all.resize((size_t)nranks * kWords);
    Member &m = g->local[0];
    m.core = core;
    m.rank = rank;
    int rc = MI355_OK;
    if (hipSetDevice(core_device(core)) != hipSuccess) rc = set_error(MI355_ERR_HIP, "hipSetDevice");
    if (!rc && hipMalloc((void **)&m.d_counts, sizeof(uint32_t) * kWords * nranks) != hipSuccess) rc = set_error(MI355_ERR_HIP, "hipMalloc");
    if (!rc && hipMalloc((void **)&m.d_send, sizeof(uint32_t) * kWords) != hipSuccess) rc = set_error(MI355_ERR_HIP, "hipMalloc");
    if (!rc) {
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        const ncclResult_t e = g_rccl.CommInitRank(&m.comm, nranks, id, rank);
        if (e != ncclSuccess) rc = rccl_fail("ncclCommInitRank", e);
    }
    if (rc) { mi355_group_destroy(g); return rc; }
    *out = g;
    return MI355_OK;
}

int mi355_group_ranks(const mi355_group *g) { return g ? g->nranks : 0; }
int mi355_group_local_members(const mi355_group *g) { return g ? (int)g->local.size() : 0; }
mi355_core *mi355_group_core(mi355_group *g, int i) {
    return g && i >= 0 && i < (int)g->local.size() ? g->local[i].core : nullptr;
}
int mi355_group_rank_of(const mi355_group *g, int i) {
    return g && i >= 0 && i < (int)g->local.size() ? g->local[i].rank : -1;
}

int mi355_group_diff_stream_batch(mi355_group *g, const void *const *d_frames, size_t stride_bytes, int nframes,
                                  void *const *d_offsets, void *const *d_xs, void *const *d_diff, size_t capacity) {
    if (int rc = check_group(g)) return rc;
    if (!d_frames || !d_offsets || !d_xs || !d_diff) return set_error(MI355_ERR_INVALID, "null argument");
    for (size_t i = 0; i < g->local.size(); i++)   // asynchronous on every member's stream: the devices run side by side
        if (int rc = mi355_diff_stream_batch(g->local[i].core, d_frames[i], stride_bytes, nframes, d_offsets[i], d_xs[i],
                                             d_diff[i], capacity))
            return rc;
    return MI355_OK;
}

int mi355_group_diff_pairs_batch(mi355_group *g, const void *const *d_cur, const void *const *d_prev,
                                 size_t stride_bytes, int nframes, void *const *d_offsets, void *const *d_xs,
                                 void *const *d_diff, size_t capacity) {
    if (int rc = check_group(g)) return rc;
    if (!d_cur || !d_prev || !d_offsets || !d_xs || !d_diff) return set_error(MI355_ERR_INVALID, "null argument");
    for (size_t i = 0; i < g->local.size(); i++)
        if (int rc = mi355_diff_pairs_batch(g->local[i].core, d_cur[i], d_prev[i], stride_bytes, nframes, d_offsets[i],
                                            d_xs[i], d_diff[i], capacity))
            return rc;
    return MI355_OK;
}

int mi355_group_synchronize(mi355_group *g) {
    if (int rc = check_group(g)) return rc;
    for (Member &m : g->local)
        if (int rc = mi355_synchronize(m.core)) return rc;
    return MI355_OK;
}

int mi355_group_gather(mi355_group *g, int root, int nframes, const void *const *d_offsets, const void *const *d_xs,
                       const void *const *d_diff, size_t member_capacity, void *d_root_offsets, void *d_root_xs,
                       void *d_root_diff, size_t root_capacity, uint64_t *h_counts) {
    if (int rc = check_group(g)) return rc;
    // Arguments only this process can judge are folded into what the ranks exchange (word 3), so that a bad call
    // on one rank fails the gather on EVERY rank instead of leaving the others in an all-gather or with sends nobody
    // receives: a root out of range, nframes < 0 or above 2^23, null arrays (bit 1), missing root buffers (bit 0).
    // The ranks' nframes travel in the same word (bits 8..31) and must agree.
    bool bad_local = root < 0 || root >= g->nranks || nframes < 0 || nframes >= (1 << 23) || !d_offsets || !d_xs || !d_diff;
    for (size_t i = 0; i < g->local.size() && !bad_local; i++) bad_local = !d_offsets[i];
    Member *rootm = bad_local ? nullptr : member_of_rank(g, root);
    const bool bad_root_args = rootm && (!d_root_offsets || (root_capacity && (!d_root_xs || !d_root_diff)));
    const int R = g->nranks;
    const uint32_t kMax = 0xFFFFFFFFu;
    // 1. what every rank must know before anything is sent: per rank {entries of its batch = offsets[nframes],
    //    entries its buffers hold, the root's capacity (the root's word only), argument check}
    int rc = MI355_OK;
    ncclResult_t ge = g_rccl.GroupStart();
    if (ge != ncclSuccess) return rccl_fail("ncclGroupStart", ge);
    for (size_t i = 0; i < g->local.size() && !rc; i++) {
        Member &m = g->local[i];
        hipStream_t s = core_stream(m.core);
        m.h_send[0] = 0;
        m.h_send[1] = member_capacity > kMax ? kMax : (uint32_t)member_capacity;
        m.h_send[2] = m.rank == root ? (root_capacity > kMax ? kMax : (uint32_t)root_capacity) : 0u;
        m.h_send[3] = (m.rank == root && bad_root_args ? 1u : 0u) | (bad_local ? 2u : (uint32_t)nframes << 8);
        hipError_t e = hipSetDevice(core_device(m.core));
        if (e == hipSuccess) e = hipMemcpyAsync(m.d_send, m.h_send, sizeof m.h_send, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && !bad_local)
            e = hipMemcpyAsync(m.d_send, (const uint32_t *)d_offsets[i] + nframes, sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { rc = set_error(MI355_ERR_HIP, hipGetErrorString(e)); break; }
        const ncclResult_t r = g_rccl.AllGather(m.d_send, m.d_counts, kWords, ncclUint32, m.comm, s);
        if (r != ncclSuccess) rc = rccl_fail("ncclAllGather", r);
    }
    ge = g_rccl.GroupEnd();   // always closed: an open group would swallow every later RCCL call of the thread
    if (!rc && ge != ncclSuccess) rc = rccl_fail("ncclGroupEnd", ge);
    if (rc) return rc;
    {
        Member &m = g->local[0];
        HIP_TRY_G(hipSetDevice(core_device(m.core)));
        HIP_TRY_G(hipMemcpyAsync(g->h_all.data(), m.d_counts, sizeof(uint32_t) * kWords * R, hipMemcpyDeviceToHost, core_stream(m.core)));
        HIP_TRY_G(hipStreamSynchronize(core_stream(m.core)));
    }
    std::vector<size_t> base(R + 1, 0);
    bool member_overflow = false, bad_root = false, bad_call = false, frames_differ = false;
    for (int r = 0; r < R; r++) {
        const uint32_t *w = &g->h_all[(size_t)r * kWords];
        g->h_counts[r] = w[0];
        base[r + 1] = base[r] + w[0];
        if (h_counts) h_counts[r] = w[0];
        member_overflow |= w[0] > w[1];
        bad_root |= (w[3] & 1u) != 0;
        bad_call |= (w[3] & 2u) != 0;
        frames_differ |= (w[3] >> 8) != (g->h_all[3] >> 8);
    }
    // the same verdict on every rank: nothing has been sent yet, nobody waits for anybody
    if (bad_call)
        return set_error(MI355_ERR_INVALID, bad_local ? "root outside [0, ranks), nframes outside [0, 2^23) or null argument"
                                                      : "another rank called the gather with invalid arguments");
    if (frames_differ) return set_error(MI355_ERR_INVALID, "the ranks disagree about nframes");
    if (bad_root) return set_error(MI355_ERR_INVALID, "the root's buffers are missing (the root is a member of some process)");
    const size_t root_cap = g->h_all[(size_t)root * kWords + 2];
    if (member_overflow)
        return set_error(MI355_ERR_INVALID, "a member's batch holds more entries than its buffers (member_capacity): "
                                            "entries beyond the capacity were dropped, there is nothing to gather from");
    if (base[R] > root_cap) return set_error(MI355_ERR_INVALID, "root capacity below the gathered total");
    // 2. index and payload travel point to point, all transfers of the step in one RCCL group
    const size_t row = (size_t)nframes + 1;
    ge = g_rccl.GroupStart();
    if (ge != ncclSuccess) return rccl_fail("ncclGroupStart", ge);
    auto step = [&](ncclResult_t r, const char *what) { if (!rc && r != ncclSuccess) rc = rccl_fail(what, r); };
    for (size_t i = 0; i < g->local.size() && !rc; i++) {
        Member &m = g->local[i];
        if (hipSetDevice(core_device(m.core)) != hipSuccess) { rc = set_error(MI355_ERR_HIP, "hipSetDevice"); break; }
        hipStream_t s = core_stream(m.core);
        if (m.rank == root) {
            for (int r = 0; r < R && !rc; r++) {
                if (r == root) continue;
                const size_t c = g->h_counts[r];
                step(g_rccl.Recv((uint32_t *)d_root_offsets + (size_t)r * row, row, ncclUint32, r, m.comm, s), "ncclRecv");
                if (c) {
                    step(g_rccl.Recv((int32_t *)d_root_xs + base[r], c, ncclInt32, r, m.comm, s), "ncclRecv");
                    step(g_rccl.Recv((uint8_t *)d_root_diff + base[r], c, ncclUint8, r, m.comm, s), "ncclRecv");
                }
            }
        } else {
            const size_t c = g->h_counts[m.rank];
            step(g_rccl.Send(d_offsets[i], row, ncclUint32, root, m.comm, s), "ncclSend");
            if (c) {
                step(g_rccl.Send(d_xs[i], c, ncclInt32, root, m.comm, s), "ncclSend");
                step(g_rccl.Send(d_diff[i], c, ncclUint8, root, m.comm, s), "ncclSend");
            }
        }
    }
    ge = g_rccl.GroupEnd();
    if (!rc && ge != ncclSuccess) rc = rccl_fail("ncclGroupEnd", ge);
    if (rc) return rc;
    // 3. the root's own part
    if (rootm) {
        const size_t i = (size_t)(rootm - g->local.data());
        HIP_TRY_G(hipSetDevice(core_device(rootm->core)));
        hipStream_t s = core_stream(rootm->core);
        const size_t c = g->h_counts[root];
        HIP_TRY_G(hipMemcpyAsync((uint32_t *)d_root_offsets + (size_t)root * row, d_offsets[i], row * sizeof(uint32_t),
                                 hipMemcpyDeviceToDevice, s));
        if (c) {
            HIP_TRY_G(hipMemcpyAsync((int32_t *)d_root_xs + base[root], d_xs[i], c * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            HIP_TRY_G(hipMemcpyAsync((uint8_t *)d_root_diff + base[root], d_diff[i], c, hipMemcpyDeviceToDevice, s));
        }
    }
    return MI355_OK;
}

}  // extern "C"
