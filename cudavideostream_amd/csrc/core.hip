// core.hip -- the C-ABI of libmi355diff.so (include/mi355diff.h): context, HBM layout, orchestration.
//
// Host-side counterpart of diff::cuda::CUDACore (reference server/src/kernels.cu:377-536).  HBM layout
// of one core (N = 3*width*height, W = ceil(N/1024) tiles, T = max_batch):
//   state     N            the client's reconstructed frame (d_current/d_previous pair of the
//                          reference collapsed into one persistent buffer)
//   in, aux, vis  N each   exec(): uploaded frame, filter scratch, visualisation frame
//   codes     ceil(T/4)*W*1024  code log written by k_diff_pack: 4 bytes per candidate lane (worst case: every lane
//                          of every frame), chunk-interleaved over tiles
//   rec       T*W*64*16    record log: 16 masked diff bytes per lane with two or more flagged bytes
//   meta      T*W*16       per (frame, tile): code position, record position, flagged bytes, candidates | multi << 16
//   groff     T*ceil(W/64)*16 (a prefix per range of 16 tiles);  totals T*8 {total, epoch} + the ticket;  offsets (T+1)*4
//   one_xs N*4, one_diff N exec(): packed output of a single frame before the D2H copies
//   hist T*256*4, thr T*4 (per frame of a filter batch), k9 9*4, heat LUT 766*3, glyph atlas
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/mi355diff.h"
#include "internal.h"

using namespace mi355;

// KiB chunks of one tile's code log: a frame appends at most 64 codes, a chunk holds 256
// (a frame's codes never straddle chunks: a chunk is left with fewer than 64 free places, i.e. more than 192 used)
static inline size_t code_chunks(size_t max_batch) { return (max_batch + 2) / 3 + 1; }

namespace {

thread_local std::string g_err;

int fail(int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                             \
    do {                                                          \
        hipError_t _e = (expr);                                   \
        if (_e != hipSuccess) return fail(MI355_ERR_HIP, #expr, _e); \
    } while (0)

}  // namespace

struct mi355_core {
    mi355_config cfg{};
    int device = 0;
    uint32_t n = 0;        // bytes per frame
    uint32_t ntiles = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    uint8_t *state = nullptr, *in = nullptr, *aux = nullptr, *vis = nullptr;
    uint4 *rec = nullptr, *meta = nullptr;
    uint32_t *codes = nullptr;
    uint32_t *groff = nullptr, *totals = nullptr;   // totals: T tagged 64-bit words + the scan kernel's ticket (2T + 2 words)
    uint64_t scan_epoch = 0;      // tag of the last k_scan_groups launch, 1 .. kEpochWrap - 1 (internal.h, next_scan_epoch)
    uint32_t *offsets = nullptr;  // T+1, used by exec()
    int32_t *one_xs = nullptr;
    uint8_t *one_diff = nullptr;
    int32_t *hist = nullptr, *thr = nullptr;
    uint32_t *red_bounds = nullptr; // mi355_red_stream_batch (cleared form): entry ranges of the frame slices
    uint8_t *gray1 = nullptr;      // fused gray+binarize chain: one gray byte per pixel of a batch, made on first use
    size_t gray1_stride = 0;
    float *k9 = nullptr;
    float *kxk = nullptr;          // mi355_conv_kxk: up to 81 taps, made on first use
    uint8_t *lut = nullptr;
    uint8_t *glyphs = nullptr;
    int nglyphs = 0, glyph_h = 0, glyph_w = 0;
    std::string charset;
    bool have_k9 = false, k9_sym = false;   // k9_sym: corners equal and edges equal, bit for bit
    size_t workspace = 0;

    uint32_t *h_count = nullptr;  // pinned, 2 x u32 (offsets[0..1] of exec)

    // pipelined per-frame path (mi355_pipe_*): a ring of slots, uploads on their own stream
    static constexpr int kMaxSlots = 8;
    struct Slot {
        uint8_t *d_in = nullptr, *d_vis = nullptr;
        uint32_t *h_count = nullptr;                   // pinned, written by k_export
        hipEvent_t uploaded = nullptr, painted = nullptr, packed = nullptr, shown = nullptr;
        bool has_vis = false;
        int64_t ticket = -1;                           // frame in flight in this slot, -1 = free
    };
    Slot slots[kMaxSlots];
    int nslots = 0;
    int64_t next_ticket = 0;
    hipStream_t up_stream = nullptr, down_stream = nullptr;

    // Pipelined batches (own stream only): the index and the expansion of batch k run on `side` beside the pack
    // kernel of batch k + 1 on the core's stream.  Two sets of logs, used alternately; set[0] is {rec, codes, meta,
    // groff, totals} above, set[1] is made when the mode is first used.
    struct LogSet {
        uint4 *rec = nullptr, *meta = nullptr;
        uint32_t *codes = nullptr, *groff = nullptr, *totals = nullptr;
        hipEvent_t packed = nullptr, expanded = nullptr;   // pack kernel done (main) / expansion done (side)
        bool in_use = false;                                // `expanded` has been recorded at least once
    };
    static constexpr int kSets = 2;   // (a third set was measured: no gain, profiles/archive/r04ay)
    LogSet set[kSets];
    int flip = 0;
    hipStream_t side = nullptr;
    hipEvent_t side_done = nullptr;   // == the `expanded` event of the last pipelined batch, or null: nothing pending
    bool pipeline_ok = true;          // false: MI355_PIPELINE=0 / MI355_OPT_PIPELINE 0, or the second set could not be allocated
    int pack_blocks_opt = -1;         // MI355_OPT_PACK_BLOCKS (-1: the default, 4 workgroups per CU)
    int median_rows = 0;              // MI355_OPT_MEDIAN_ROWS (0: chosen per launch)
    uint32_t k1_blocks = 0;           // pipelined batches: workgroups of the pack kernel (0 = one tile per wave)
    uint32_t cu_count = 0;            // compute units of the device
    // pipelined batches packed by TWO launches (tiles [0, split) on the core's stream, the rest on `main2`): the two chains
    // of pack kernels drift apart, each one's kernel boundary (L2 write-back, event packets: 21-25 us) falls into the other's
    // kernel.  split_pct = 0: one launch.
    // (three or four launches were measured: much slower, profiles/archive/r04ah)
    int split_pct = 50;               // MI355_OPT_SPLIT_PCT
    hipStream_t main2 = nullptr;
    hipEvent_t packed2[kSets] = {};
    // Adaptive overlap: a batch whose expansion is longer than its pack kernel (dense input: a scene change, the synthetic
    // worst cases) loses by running beside the next batch's pack kernel (S0 pairs 0.313 ms one after the other, 0.35
    // overlapped).  The batch total (offsets[nframes]) of every own-stream batch is copied to pinned host memory behind its
    // expansion; the next calls look at the latest total that HAS ARRIVED (a word of pinned memory the index kernel stores, no
    // waiting) and run one batch after the other while more than dense_pct per cent of the bytes changed.  Results never depend
    // on it, only the schedule.
    uint64_t *h_tot = nullptr;        // pinned: {entries of the latest own-stream batch whose index has run, its frames << 32},
                                      // stored by that batch's index kernel itself (k_scan_groups, `note`)
    int dense_pct = 40;               // MI355_OPT_DENSE_PCT (0: never switch)
    bool dense = false;               // what the latest total that has arrived said
    bool filter_since_batch = false;  // a frame filter ran on this core since the last batch (use_device_filter)
    bool chain_hint = true;           // MI355_OPT_CHAIN_HINT 0: batches are overlapped regardless
    hipEvent_t fork[kSets] = {};          // recorded on the core's stream in front of a batch's first pack launch: the other parts wait for it
    int parts_pending = -1;           // log set of the last batch whose parts the core's stream has not waited for (-1: none)

    // timing: ring of event sets {before pack, after pack, before scan, after scan, after expand, after the second pack
    // launch of a split batch}, harvested lazily so that timed batches still queue back to back
    static constexpr int kEvRing = 32, kEvPer = 6;
    bool timing = false;
    hipEvent_t ev[kEvRing][kEvPer] = {};
    bool ev_split[kEvRing] = {};
    int ev_head = 0, ev_count = 0;  // oldest pending slot, number pending
    double ms_pack = 0, ms_scan = 0, ms_expand = 0, ms_total = 0;
    int launches = 0;
};

namespace mi355 {
int set_error(int code, const char *what) { return fail(code, what); }
hipStream_t core_stream(::mi355_core *c) {   // for work other translation units enqueue: after the last pipelined batch
    if (c->side_done && hipSetDevice(c->device) == hipSuccess && hipStreamWaitEvent(c->stream, c->side_done, 0) == hipSuccess)
        c->side_done = nullptr;
    return c->stream;
}
int core_device(const ::mi355_core *c) { return c->device; }
}  // namespace mi355

namespace {

// A stream of the core.  MI355_FLAG_OWN_QUEUES: in the least stream-priority class, whose hardware queues no framework's
// stream pool shares (include/mi355diff.h); else the default class.
hipError_t make_stream(const mi355_core *c, hipStream_t *s) {
    if (c->cfg.flags & MI355_FLAG_OWN_QUEUES) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
        (void)hipGetLastError();
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

template <class T>
int dev_alloc(mi355_core *c, T **p, size_t count) {
    const size_t bytes = count * sizeof(T);
    HIP_TRY(hipMalloc((void **)p, bytes ? bytes : 16));
    c->workspace += bytes;
    return MI355_OK;
}

// Every entry point starts here.  join: whatever the entry point enqueues on the core's stream (or waits for) comes
// after the expansion of the last pipelined batch, which runs on the side stream; only the pipelined batch path
// itself passes false (it orders its own kernels with events).
// parts: a pipelined batch is packed by launches on SEVERAL streams (run_batch); whatever follows on the core's stream must
// not overtake the parts that run elsewhere -- a frame filter that rewrites the buffer the batch is still reading, say.
// Only the next pipelined batch passes false here too (its own parts queue up behind the last batch's on their streams).
int use_device(mi355_core *c, bool join = true, bool parts = true) {
    HIP_TRY(hipSetDevice(c->device));
    if (join && c->side_done) {
        HIP_TRY(hipStreamWaitEvent(c->stream, c->side_done, 0));   // (the expansion waited for every part)
        c->side_done = nullptr;
        c->parts_pending = -1;
    }
    if (parts && c->parts_pending >= 0) {
        HIP_TRY(hipStreamWaitEvent(c->stream, c->packed2[c->parts_pending], 0));
        c->parts_pending = -1;
    }
    return MI355_OK;
}

// A frame filter takes frames, not packed streams: it does not wait for a pipelined batch that is still expanding on the
// side stream.  It leaves a hint, though: a caller that alternates filters and batches (the server's visualiser or noise
// filter in front of every diff: BASELINE configs 3 and 4) gains nothing from a batch's expansion running beside the next
// filter -- both are bound by the memory system -- and loses the pipelined batch's smaller pack grid and stream hops:
// 4.6 us per frame one after the other, 5.0-5.2 overlapped (config 3, profiles/archive/r04bd).  The next batch therefore runs one
// kernel after the other on the core's stream (MI355_OPT_CHAIN_HINT 0: ignore the hint).
int use_device_filter(mi355_core *c) {
    c->filter_since_batch = c->chain_hint;
    return use_device(c, false);
}

// Fold pending event sets into the sums: all of them (blocking) or only until `keep` remain.
int harvest_timing(mi355_core *c, int keep = 0) {
    while (c->ev_count > keep) {
        hipEvent_t *e = c->ev[c->ev_head];
        HIP_TRY(hipEventSynchronize(e[4]));
        float a = 0, b = 0, d = 0, t = 0;
        HIP_TRY(hipEventElapsedTime(&a, e[0], e[1]));
        if (c->ev_split[c->ev_head]) {   // two launches on two streams: the later end counts
            float a2 = 0;
            HIP_TRY(hipEventElapsedTime(&a2, e[0], e[5]));
            a = a2 > a ? a2 : a;
        }
        HIP_TRY(hipEventElapsedTime(&b, e[2], e[3]));
        HIP_TRY(hipEventElapsedTime(&d, e[3], e[4]));
        HIP_TRY(hipEventElapsedTime(&t, e[0], e[4]));   // the batch's way through the path (batches overlap when pipelined)
        c->ms_pack += a;
        c->ms_scan += b;
        c->ms_expand += d;
        c->ms_total += t;
        c->launches += 1;
        c->ev_head = (c->ev_head + 1) % mi355_core::kEvRing;
        c->ev_count -= 1;
    }
    return MI355_OK;
}

// tests/heat_map_benchmark/cpu.cu:19-27 for d = 0..765, stored B,G,R (cpu.cu:62-64).
void build_heat_lut(uint8_t *lut) {
    for (int diff = 0; diff <= 765; diff++) {
        float diff1 = diff / (255.0 * 2.0);
        double r = sin(M_PI * diff1 - M_PI / 2.0) * 255.0;
        double g = sin(M_PI * diff1) * 255.0;
        double b = sin(M_PI * diff1 + M_PI / 2.0) * 255.0;
        r = r > 0.0 ? r : 0.0; r = r < 255.0 ? r : 255.0;
        g = g > 0.0 ? g : 0.0; g = g < 255.0 ? g : 255.0;
        b = b > 0.0 ? b : 0.0; b = b < 255.0 ? b : 255.0;
        lut[diff * 3 + 0] = (uint8_t)(int)b;
        lut[diff * 3 + 1] = (uint8_t)(int)g;
        lut[diff * 3 + 2] = (uint8_t)(int)r;
    }
}

// Scratch of the fused gray+binarize chain (one byte per pixel, max_batch frames): made when first needed.
int need_gray1(mi355_core *c) {
    if (c->gray1 || c->n == 0) return MI355_OK;
    c->gray1_stride = ((size_t)c->n / 3 + 15) & ~(size_t)15;
    return dev_alloc(c, &c->gray1, c->gray1_stride * (size_t)c->cfg.max_batch);
}

// Every frame total of both log sets back to "never written" (the launch tags wrap, or a test moves them).  The caller has
// synchronised the core's streams.  The first version cleared with plain hipMemset calls -- served by the null stream / the
// runtime's own fill path, not by the core's streams -- and tests/soak_chain.py (round 6, the tag moved at random) then stopped
// in the index kernel's reader, waiting for a total that had been published: 3 runs of 3; with the clears enqueued on the core's
// stream and waited for, 5 of 5 ran to the end (tools/exp/r06s.sh).  What exactly let the published word be lost was not
// established (a core created while the null stream is busy behaves the same with either form: test_core_created_while_the_
// null_stream_is_busy); every clear of the core's own buffers goes the ordered way since.
int clear_totals(mi355_core *c) {
    const size_t bytes = 2 * (size_t)c->cfg.max_batch * sizeof(uint32_t);   // (the ticket behind them is 0 between launches)
    HIP_TRY(hipMemsetAsync(c->totals, 0, bytes, c->stream));
    if (c->set[1].totals) HIP_TRY(hipMemsetAsync(c->set[1].totals, 0, bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MI355_OK;
}

// Slice bounds of the cleared red map (mi355_red_stream_batch, clear != 0), max_batch frames: made when first needed.
int need_red_bounds(mi355_core *c) {
    if (c->red_bounds || c->n == 0) return MI355_OK;
    return dev_alloc(c, &c->red_bounds, (size_t)c->cfg.max_batch * red_bounds_per_frame(c->n));
}

// The second set of logs and the side stream of the pipelined mode, made when it is first used (or by mi355_prepare).
int setup_pipeline(mi355_core *c) {
    if (c->side || !c->pipeline_ok) return MI355_OK;
    // Pipelined batches: the pack kernel on 4 workgroups per CU (16 waves: half of them walk a second tile) instead of
    // one tile per wave (6 per CU).  It is bound by the memory system and does not need its occupancy (profiles/README.md,
    // round 1), while the expansion of the batch before, which shares the chip with it, lives on the wave slots and
    // registers that are left: 0.503-0.507 -> 0.487-0.497 ms per batch on the faster boxes, +-1 % on the slower ones
    // (profiles/archive/r04p, r04q, r04v).  MI355_OPT_PACK_BLOCKS overrides (0 = one tile per wave).
    if (c->pack_blocks_opt >= 0) {
        c->k1_blocks = (uint32_t)c->pack_blocks_opt;
    } else {
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
            c->k1_blocks = 4u * (uint32_t)prop.multiProcessorCount;
    }
    const size_t T = (size_t)c->cfg.max_batch, W = c->ntiles;
    mi355_core::LogSet &s0 = c->set[0], &s1 = c->set[1];
    s0.rec = c->rec; s0.codes = c->codes; s0.meta = c->meta; s0.groff = c->groff; s0.totals = c->totals;
    bool ok = true;
    ok = ok && hipMalloc((void **)&s1.rec, T * W * 1024) == hipSuccess;
    ok = ok && hipMalloc((void **)&s1.codes, code_chunks(T) * W * 1024) == hipSuccess;
    ok = ok && hipMalloc((void **)&s1.meta, T * W * 16) == hipSuccess;
    ok = ok && hipMalloc((void **)&s1.groff, T * expand_groups(c->ntiles) * 16) == hipSuccess;
    ok = ok && hipMalloc((void **)&s1.totals, (2 * T + 2) * sizeof(uint32_t)) == hipSuccess;
    // (through the core's stream, and waited for -- clear_totals)
    ok = ok && hipMemsetAsync(s1.totals, 0, (2 * T + 2) * sizeof(uint32_t), c->stream) == hipSuccess &&   // tag 0 = never written; the ticket
         hipStreamSynchronize(c->stream) == hipSuccess;
    ok = ok && make_stream(c, &c->side) == hipSuccess;
    ok = ok && make_stream(c, &c->main2) == hipSuccess;
    for (int i = 0; i < mi355_core::kSets && ok; i++) {
        // device-scope release: these events only order kernels of this device against each other.  An event's default
        // is a SYSTEM-scope fence when it is recorded (caches written back and invalidated for the host's benefit),
        // which every batch paid twice on the core's stream between two pack kernels
        const unsigned flags = hipEventDisableTiming | hipEventReleaseToDevice;
        ok = hipEventCreateWithFlags(&c->set[i].packed, flags) == hipSuccess &&
             hipEventCreateWithFlags(&c->set[i].expanded, flags) == hipSuccess &&
             hipEventCreateWithFlags(&c->packed2[i], flags) == hipSuccess &&
             hipEventCreateWithFlags(&c->fork[i], flags) == hipSuccess;
    }
    if (!ok) {   // not an error: the batches then run one after the other, as with a caller's stream
        (void)hipGetLastError();
        void *ptrs[] = {s1.rec, s1.codes, s1.meta, s1.groff, s1.totals};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        s1.rec = nullptr; s1.meta = nullptr; s1.codes = s1.groff = s1.totals = nullptr;
        if (c->side) { (void)hipStreamDestroy(c->side); c->side = nullptr; }
        if (c->main2) { (void)hipStreamDestroy(c->main2); c->main2 = nullptr; }
        // ... and what a later attempt (MI355_OPT_PIPELINE 1 switches the mode on again) would otherwise overwrite
        for (int i = 0; i < mi355_core::kSets; i++) {
            hipEvent_t *evs[] = {&c->set[i].packed, &c->set[i].expanded, &c->packed2[i], &c->fork[i]};
            for (hipEvent_t *e : evs) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        }
        c->pipeline_ok = false;
        return MI355_OK;
    }
    c->workspace += T * W * 1024 + code_chunks(T) * W * 1024 + T * W * 16 + T * expand_groups(c->ntiles) * 16 + (2 * T + 2) * 4;
    return MI355_OK;
}

// d_wire != nullptr: the expander writes the sender's byte stream (capacity in bytes) instead of d_xs/d_diff.
// pipelined: a public batch entry point on the core's OWN stream -- the index and the expansion run on the side
// stream beside the next batch's pack kernel (which needs only the state, carried on the core's stream, and a free
// set of logs); completion is what mi355_synchronize / any other entry point waits for (use_device joins).  With a
// caller's stream (mi355_set_stream) everything stays on that stream, in order: a caller who enqueues its own
// consumers there must find the batch complete.
int run_batch(mi355_core *c, bool pair, const void *d_cur, const void *d_prev, size_t stride,
              int nframes, void *d_offsets, void *d_xs, void *d_diff, size_t capacity,
              void *d_wire = nullptr, bool pipelined = false) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (nframes < 0 || nframes > c->cfg.max_batch)
        return fail(MI355_ERR_INVALID, "nframes outside [0, max_batch]");
    if (!d_offsets) return fail(MI355_ERR_INVALID, "null d_offsets");
    if (nframes > 0 && c->n > 0 && (!d_cur || (pair && !d_prev)))
        return fail(MI355_ERR_INVALID, "null frame pointer");
    if (nframes > 0 && stride < c->n) return fail(MI355_ERR_INVALID, "stride_bytes < frame bytes");
    if (capacity > 0 && !d_wire && (!d_xs || !d_diff)) return fail(MI355_ERR_INVALID, "null output pointer");
    pipelined = pipelined && c->stream == c->own_stream && c->pipeline_ok && nframes > 0 && c->n > 0;
    const bool own = pipelined;   // an own-stream batch (its total is recorded for the next decisions)
    if (c->filter_since_batch) pipelined = false;   // a filter / batch chain: one kernel after the other (use_device_filter)
    c->filter_since_batch = false;
    if (pipelined) {
        if (int rc = use_device(c, false, false)) return rc;
        if (int rc = setup_pipeline(c)) return rc;
        pipelined = c->pipeline_ok;
    }
    if (pipelined && c->h_tot && c->dense_pct > 0 && c->dense_pct < 100) {
        // the latest batch total that has arrived: dense input -> this batch runs after the expansion of the one before
        const uint64_t note = __atomic_load_n(c->h_tot, __ATOMIC_RELAXED);   // one 64-bit word: never torn
        if (note >> 32) c->dense = (note & 0xffffffffull) * 100u > (uint64_t)c->dense_pct * (note >> 32) * c->n;
        if (c->dense) pipelined = false;   // (no total has arrived yet: what the last one said still holds)
    }
    if (!pipelined)
        if (int rc = use_device(c)) return rc;
    if (nframes == 0 || c->n == 0) {
        HIP_TRY(hipMemsetAsync(d_offsets, 0, sizeof(uint32_t) * ((size_t)nframes + 1), c->stream));
        if (d_wire) {   // empty frames still have their headers {n = 0}
            const size_t head = 4 * (size_t)nframes < capacity ? 4 * (size_t)nframes : capacity & ~(size_t)3;
            if (head) HIP_TRY(hipMemsetAsync(d_wire, 0, head, c->stream));
        }
        return MI355_OK;
    }
    hipEvent_t *tev = nullptr;
    if (c->timing) {
        if (int rc = harvest_timing(c, mi355_core::kEvRing - 1)) return rc;
        tev = c->ev[(c->ev_head + c->ev_count) % mi355_core::kEvRing];
    }
    // the set of logs this batch writes; a pipelined batch may only start once the expansion that last read the
    // set (two batches ago) is done
    mi355_core::LogSet one;
    one.rec = c->rec; one.codes = c->codes; one.meta = c->meta; one.groff = c->groff; one.totals = c->totals;
    mi355_core::LogSet &ls = pipelined ? c->set[c->flip] : one;
    hipStream_t tail = pipelined ? c->side : c->stream;   // where the index and the expansion run
    if (pipelined && ls.in_use) HIP_TRY(hipStreamWaitEvent(c->stream, ls.expanded, 0));
    if (tev) HIP_TRY(hipEventRecord(tev[0], c->stream));
    PackArgs a{};
    a.cur = (const uint8_t *)d_cur;
    a.prev = (const uint8_t *)d_prev;
    a.state = c->state;
    a.stride = stride;
    a.n = c->n;
    a.nframes = nframes;
    a.thr = c->cfg.threshold;
    a.ntiles = c->ntiles;
    a.tile_begin = 0;
    a.tile_end = c->ntiles;
    a.rec = ls.rec;
    a.codes = ls.codes;
    a.codes_bytes = (uint32_t)(code_chunks((size_t)c->cfg.max_batch) * c->ntiles * 1024u);
    a.meta = ls.meta;
    a.rec_bytes = (uint32_t)((size_t)c->cfg.max_batch * c->ntiles * 1024u);
    a.meta_bytes = (uint32_t)((size_t)c->cfg.max_batch * c->ntiles * 16u);
    // the vector path of the pack kernel: 16-byte aligned operands, and a group of four frames within reach of one
    // buffer descriptor's 32-bit offsets (diff_pack.hip, Group::load_desc)
    const bool aligned = (((uintptr_t)d_cur | (uintptr_t)d_prev | stride) & 15u) == 0 && 3 * (uint64_t)stride + c->n < (1ull << 32);
    // Pair mode: do the two operands share a frame (or part of one), i.e. does prev + j * stride come within a frame of
    // cur + i * stride for some i, j < T?  Pairs of consecutive frames do (cur of one pair is prev of the next: the second
    // read hits in the caches); the pairs of a round-robin shard do not, and are read with non-temporal loads like the
    // frames of a stream (diff_pack.hip, Group::load_desc).
    bool pair_once = false;
    if (pair) {
        const int64_t d = (int64_t)((intptr_t)d_prev - (intptr_t)d_cur), st = (int64_t)stride, T = nframes;
        bool shared = false;
        const int64_t k0 = d / st;
        for (int64_t k = k0 - 1; k <= k0 + 1; k++)
            if (k > -T && k < T && (d - k * st < 0 ? k * st - d : d - k * st) < (int64_t)c->n) shared = true;
        pair_once = !shared;
    }
    const bool split = pipelined && c->split_pct && c->main2 && c->ntiles >= 64;
    if (split) {
        // tiles [0, cut) on the core's stream, the rest on a stream of its own (which also has to see the log set free
        // and everything the core's stream holds so far: a filter that is still writing the frames this batch reads,
        // the upload of the state); each part takes its share of the pipelined grid
        const uint32_t cut = (uint32_t)((uint64_t)c->ntiles * (uint32_t)c->split_pct / 100u) & ~3u;
        uint32_t blocks0 = 0, blocks1 = 0;
        if (c->k1_blocks) {
            blocks0 = (uint32_t)((uint64_t)c->k1_blocks * cut / c->ntiles);
            if (blocks0 == 0) blocks0 = 1;
            blocks1 = c->k1_blocks > blocks0 ? c->k1_blocks - blocks0 : 1u;
        }
        HIP_TRY(hipEventRecord(c->fork[c->flip], c->stream));
        PackArgs a0 = a, a1 = a;
        a0.tile_end = cut;
        a1.tile_begin = cut;
        HIP_TRY(launch_diff_pack(a0, pair, aligned, pair_once, blocks0, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->main2, c->fork[c->flip], 0));
        if (ls.in_use) HIP_TRY(hipStreamWaitEvent(c->main2, ls.expanded, 0));
        HIP_TRY(launch_diff_pack(a1, pair, aligned, pair_once, blocks1, c->main2));
        HIP_TRY(hipEventRecord(c->packed2[c->flip], c->main2));
        HIP_TRY(hipStreamWaitEvent(tail, c->packed2[c->flip], 0));
        if (tev) HIP_TRY(hipEventRecord(tev[5], c->main2));   // the pack "kernel" of a split batch ends when BOTH parts have
        c->parts_pending = c->flip;
    } else {
        HIP_TRY(launch_diff_pack(a, pair, aligned, pair_once, pipelined ? c->k1_blocks : 0u, c->stream));
    }
    if (tev) {
        HIP_TRY(hipEventRecord(tev[1], c->stream));
        c->ev_split[(c->ev_head + c->ev_count) % mi355_core::kEvRing] = split;
    }
    // The index kernel is short (12 us alone) and gates the expansion; it runs in front of it on the side stream (on the
    // core's stream, between two pack kernels, was measured: worse, profiles/archive/r04a).
    if (pipelined) {
        HIP_TRY(hipEventRecord(ls.packed, c->stream));
        HIP_TRY(hipStreamWaitEvent(tail, ls.packed, 0));
    }
    if (tev) HIP_TRY(hipEventRecord(tev[2], tail));
    if (next_scan_epoch(c->scan_epoch)) {
        // the launch tags have wrapped (2^33 launches): every total any launch of this core has written goes, behind a
        // synchronisation, before a tag is used a second time -- a slot can then never hold a stale word with a current tag
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->side) HIP_TRY(hipStreamSynchronize(c->side));
        if (c->main2) HIP_TRY(hipStreamSynchronize(c->main2));
        if (int rc = clear_totals(c)) return rc;
    }
    // (an own-stream batch leaves its total in pinned memory for the next calls' decisions -- stored by the index kernel
    // itself: a copy + an event behind every batch cost config 3's chain 4 %, the event's system-scope fence included)
    uint64_t *const note = own && c->h_tot && c->pipeline_ok && c->dense_pct > 0 && c->dense_pct < 100 ? c->h_tot : nullptr;
    HIP_TRY(launch_scan(ls.meta, ls.groff, (uint64_t *)ls.totals, c->ntiles, nframes, (uint32_t *)d_offsets,
                        ls.totals + 2 * (size_t)c->cfg.max_batch, c->scan_epoch, note, tail));
    if (tev) HIP_TRY(hipEventRecord(tev[3], tail));
    ExpandArgs g{};
    g.rec = ls.rec;
    g.codes = ls.codes;
    g.meta = ls.meta;
    g.roff = ls.groff;
    g.offsets = (const uint32_t *)d_offsets;
    g.ntiles = c->ntiles;
    g.codes_bytes = a.codes_bytes;
    g.rec_bytes = a.rec_bytes;
    g.out_xs = (int32_t *)d_xs;
    g.out_diff = (uint8_t *)d_diff;
    g.wire = (uint8_t *)d_wire;
    g.capacity = capacity;
    HIP_TRY(launch_expand(g, nframes, tail));
    if (tev) {
        HIP_TRY(hipEventRecord(tev[4], tail));
        c->ev_count += 1;
    }
    if (pipelined) {
        HIP_TRY(hipEventRecord(ls.expanded, tail));
        ls.in_use = true;
        c->side_done = ls.expanded;
        c->flip ^= 1;
    }
    return MI355_OK;
}

}  // namespace

extern "C" {

const char *mi355_last_error(void) { return g_err.c_str(); }

int mi355_abi_version(void) { return MI355_ABI_VERSION; }

int mi355_create(const mi355_config *cfg, mi355_core **out) {
    if (!cfg || !out) return fail(MI355_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->width < 0 || cfg->height < 0) return fail(MI355_ERR_INVALID, "negative frame size");
    // |df| of two bytes is at most 255: a threshold of 255 flags nothing; larger or negative values are refused
    if (cfg->threshold < 0 || cfg->threshold > 255) return fail(MI355_ERR_INVALID, "threshold outside 0..255");
    if (cfg->max_batch < 1) return fail(MI355_ERR_INVALID, "max_batch < 1");
    if (cfg->visualizer < 0 || cfg->visualizer > 5) return fail(MI355_ERR_INVALID, "unknown visualizer");
    if (cfg->flags & ~MI355_FLAG_OWN_QUEUES) return fail(MI355_ERR_INVALID, "flags: unknown bit (MI355_FLAG_OWN_QUEUES is the only flag of this version of the library)");
    const uint64_t n64 = 3ull * (uint64_t)cfg->width * (uint64_t)cfg->height;
    if (n64 >= (1ull << 31)) return fail(MI355_ERR_INVALID, "frame larger than 2 GiB");
    // byte indices are int32 and batch offsets uint32 (the reference's h_xs / h_pos types)
    // (whole tiles: the record log, max_batch * ceil(N / 1024) KiB, is addressed with 32-bit byte offsets)
    // ... and so is the code log, code_chunks(max_batch) * ceil(N / 1024) KiB, which is the larger one for max_batch = 1
    // (two chunks per tile: a frame's codes never straddle a chunk)
    {
        const uint64_t tiles = (n64 + kTileBytes - 1) / kTileBytes, mb = (uint64_t)cfg->max_batch;
        const uint64_t chunks = mb > code_chunks((size_t)mb) ? mb : (uint64_t)code_chunks((size_t)mb);
        if (tiles * kTileBytes * chunks >= (1ull << 32))
            return fail(MI355_ERR_INVALID, "max_batch * frame bytes must stay below 2^32");
    }

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(MI355_ERR_HIP, "no HIP device available (libmi355diff has no CPU fallback)", e);
    mi355_core *c = new (std::nothrow) mi355_core;
    if (!c) return fail(MI355_ERR_INVALID, "out of host memory");
    c->cfg = *cfg;
    // the ONE environment variable the library reads (include/mi355diff.h, "Options")
    if (const char *v = getenv("MI355_PIPELINE")) c->pipeline_ok = v[0] != '0';
    if (cfg->device >= 0) c->device = cfg->device;
    else if ((e = hipGetDevice(&c->device)) != hipSuccess) { delete c; return fail(MI355_ERR_HIP, "hipGetDevice", e); }
    if (c->device >= ndev) { delete c; return fail(MI355_ERR_INVALID, "device ordinal out of range"); }

    c->n = (uint32_t)n64;
    c->ntiles = (c->n + kTileBytes - 1) / kTileBytes;
    {
        hipDeviceProp_t prop{};
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->cu_count = (uint32_t)prop.multiProcessorCount;
    }
    const size_t T = (size_t)cfg->max_batch, W = c->ntiles, N = c->n;

    int rc = use_device(c);
    if (!rc) { e = make_stream(c, &c->own_stream); if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipStreamCreate", e); }
    c->stream = c->own_stream;
    for (int i = 0; i < mi355_core::kEvRing * mi355_core::kEvPer && !rc; i++) { e = hipEventCreateWithFlags(&c->ev[i / mi355_core::kEvPer][i % mi355_core::kEvPer], hipEventDisableSystemFence); if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipEventCreate", e); }   // timing events: no system-scope fence (hip_runtime_api.h: "can improve the accuracy of timing measurements by avoiding the cost of cache writeback and invalidation")
    if (!rc && (e = init_gray_table()) != hipSuccess) rc = fail(MI355_ERR_HIP, "init_gray_table", e);
    if (!rc) rc = dev_alloc(c, &c->state, N + 16);
    if (!rc) rc = dev_alloc(c, &c->in, N + 16);
    if (!rc) rc = dev_alloc(c, &c->aux, N + 16);
    if (!rc) rc = dev_alloc(c, &c->vis, N + 16);
    if (!rc) rc = dev_alloc(c, &c->rec, T * W * 64);
    if (!rc) rc = dev_alloc(c, &c->codes, code_chunks(T) * W * 256);
    if (!rc) rc = dev_alloc(c, &c->meta, T * W);
    if (!rc) rc = dev_alloc(c, &c->groff, T * expand_groups(c->ntiles) * 4);   // one prefix per range of 16 tiles
    if (!rc) rc = dev_alloc(c, &c->totals, 2 * T + 2);   // T x {total, epoch} + the scan kernel's ticket counter
    if (!rc) { e = hipMemsetAsync(c->totals, 0, (2 * T + 2) * sizeof(uint32_t), c->own_stream); if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipMemset", e); }
    if (!rc) rc = dev_alloc(c, &c->offsets, T + 1);
    if (!rc) rc = dev_alloc(c, &c->one_xs, N + 4);
    if (!rc) rc = dev_alloc(c, &c->one_diff, N + 16);
    if (!rc) rc = dev_alloc(c, &c->hist, 256 * T);
    if (!rc) rc = dev_alloc(c, &c->thr, T);
    if (!rc) rc = dev_alloc(c, &c->k9, 9);
    if (!rc) rc = dev_alloc(c, &c->lut, 768 * 3);
    if (!rc) { e = hipHostMalloc((void **)&c->h_count, 2 * sizeof(uint32_t), hipHostMallocDefault); if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipHostMalloc", e); }
    if (!rc) { e = hipHostMalloc((void **)&c->h_tot, sizeof(uint64_t), hipHostMallocDefault); if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipHostMalloc", e); else *c->h_tot = 0; }
    if (!rc) { e = hipMemsetAsync(c->state, 0, N + 16, c->own_stream); if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipMemset", e); }
    // (the clears above went through the core's own stream -- clear_totals -- and are complete before the core is handed out)
    if (!rc) { e = hipStreamSynchronize(c->own_stream); if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipStreamSynchronize", e); }
    if (!rc) {
        uint8_t lut[768 * 3] = {0};
        build_heat_lut(lut);
        e = hipMemcpy(c->lut, lut, sizeof lut, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(MI355_ERR_HIP, "hipMemcpy(lut)", e);
    }
    // what a per-frame server's first exec() would otherwise allocate (kernels.cu:493-498: visualiser 5)
    if (!rc && cfg->visualizer == MI355_VIS_BINARIZE) rc = need_gray1(c);
    if (rc) { mi355_destroy(c); return rc; }
    *out = c;
    return MI355_OK;
}

void mi355_destroy(mi355_core *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->nslots) (void)mi355_pipe_close(c);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    {
        mi355_core::LogSet &s1 = c->set[1];
        void *more[] = {s1.rec, s1.codes, s1.meta, s1.groff, s1.totals};
        for (void *p : more) if (p) (void)hipFree(p);
        for (auto &ls : c->set) {
            if (ls.packed) (void)hipEventDestroy(ls.packed);
            if (ls.expanded) (void)hipEventDestroy(ls.expanded);
        }
        if (c->side) (void)hipStreamDestroy(c->side);
        if (c->main2) { (void)hipStreamSynchronize(c->main2); (void)hipStreamDestroy(c->main2); }
        for (auto &e : c->packed2) if (e) (void)hipEventDestroy(e);
        for (auto &e : c->fork) if (e) (void)hipEventDestroy(e);
        if (c->h_tot) (void)hipHostFree(c->h_tot);
    }
    void *ptrs[] = {c->state, c->in, c->aux, c->vis, c->rec, c->codes, c->meta, c->groff, c->totals, c->offsets, c->one_xs, c->one_diff, c->hist, c->thr, c->k9,
                    c->lut, c->glyphs, c->kxk, c->gray1, c->red_bounds};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->h_count) (void)hipHostFree(c->h_count);
    for (auto &slot : c->ev) for (auto &ev : slot) if (ev) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

static int warm_exec(mi355_core *c);   // below, beside mi355_exec

int mi355_prepare(mi355_core *c, unsigned what) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (what & ~MI355_PREPARE_ALL) return fail(MI355_ERR_INVALID, "mi355_prepare: unknown bit in `what`");
    if (int rc = use_device(c)) return rc;
    if ((what & MI355_PREPARE_BATCHES) && c->n > 0) {
        if (int rc = setup_pipeline(c)) return rc;   // (a core that cannot have its second set stays sequential: not an error)
    }
    if (what & MI355_PREPARE_GRAY_CHAIN)
        if (int rc = need_gray1(c)) return rc;
    if (what & MI355_PREPARE_RED_CLEAR)
        if (int rc = need_red_bounds(c)) return rc;
    if ((what & MI355_PREPARE_CONV_KXK) && !c->kxk)
        if (int rc = dev_alloc(c, &c->kxk, 81)) return rc;
    if ((what & MI355_PREPARE_EXEC) && c->n > 0 && c->nslots == 0)
        if (int rc = warm_exec(c)) return rc;
    HIP_TRY(hipDeviceSynchronize());   // the memsets of the new buffers
    return MI355_OK;
}

size_t mi355_frame_bytes(const mi355_core *c) { return c ? c->n : 0; }
size_t mi355_workspace_bytes(const mi355_core *c) { return c ? c->workspace : 0; }

int mi355_set_stream(mi355_core *c, void *hip_stream) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    if (int rc = harvest_timing(c)) return rc;
    // every batch shares one workspace (log, meta, scans, state): work queued on the old stream must be done
    // before anything is enqueued on the new one
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)hip_stream;  // NULL is the (legacy) default stream, a valid choice
    return MI355_OK;
}

int mi355_use_own_stream(mi355_core *c) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    if (int rc = harvest_timing(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));   // see mi355_set_stream
    c->stream = c->own_stream;
    return MI355_OK;
}

int mi355_synchronize(mi355_core *c) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MI355_OK;
}

// Options: the schedule of the own-stream batches, never a result.  Changing one first completes what is queued.
int mi355_set_option(mi355_core *c, int option, int value) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    switch (option) {
        case MI355_OPT_PIPELINE:
            if (value != 0 && value != 1) return fail(MI355_ERR_INVALID, "MI355_OPT_PIPELINE: 0 or 1");
            // (a core whose second log set could not be allocated stays sequential: setup_pipeline says so by itself)
            c->pipeline_ok = value == 1;
            return MI355_OK;
        case MI355_OPT_SPLIT_PCT:
            if (value != 0 && (value < 5 || value > 95)) return fail(MI355_ERR_INVALID, "MI355_OPT_SPLIT_PCT: 0 or 5..95");
            c->split_pct = value;
            return MI355_OK;
        case MI355_OPT_DENSE_PCT:
            if (value < 0 || value > 100) return fail(MI355_ERR_INVALID, "MI355_OPT_DENSE_PCT: 0..100");
            c->dense_pct = value;
            c->dense = false;
            return MI355_OK;
        case MI355_OPT_CHAIN_HINT:
            if (value != 0 && value != 1) return fail(MI355_ERR_INVALID, "MI355_OPT_CHAIN_HINT: 0 or 1");
            c->chain_hint = value == 1;
            c->filter_since_batch = false;
            return MI355_OK;
        case MI355_OPT_PACK_BLOCKS:
            if (value < -1 || value > (1 << 20)) return fail(MI355_ERR_INVALID, "MI355_OPT_PACK_BLOCKS: -1 (default), 0 (one tile per wave) or a workgroup count");
            c->pack_blocks_opt = value;
            if (c->side) {   // the pipelined mode is already set up: takes effect with the next batch
                if (value >= 0) c->k1_blocks = (uint32_t)value;
                else {
                    hipDeviceProp_t prop{};
                    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
                    c->k1_blocks = 4u * (uint32_t)prop.multiProcessorCount;
                }
            }
            return MI355_OK;
        case MI355_OPT_MEDIAN_ROWS:
            if (value < 0 || value > 60 || value % 5) return fail(MI355_ERR_INVALID, "MI355_OPT_MEDIAN_ROWS: 0 (default) or 5, 10, .. 60");
            c->median_rows = value;
            return MI355_OK;
        case MI355_OPT_SCAN_EPOCH_LEFT:   // tests: the index kernel's launch tag this many launches before its wrap
            if (value < 1 || value > (1 << 30)) return fail(MI355_ERR_INVALID, "MI355_OPT_SCAN_EPOCH_LEFT: 1..2^30");
            if (c->side) HIP_TRY(hipStreamSynchronize(c->side));
            if (c->main2) HIP_TRY(hipStreamSynchronize(c->main2));
            // every total goes with the jump: a tag set BACK (the option used twice before a wrap) would otherwise meet slots that
            // still carry that very tag from an earlier launch, and the index kernel would take their stale totals for this launch's
            // (found by tests/soak_chain.py in round 6: offsets of garbage, a red-map kernel walking 2^32 entries)
            if (int rc = clear_totals(c)) return rc;
            c->scan_epoch = kEpochWrap - 1 - (uint64_t)value;
            return MI355_OK;
        default: return fail(MI355_ERR_INVALID, "unknown option");
    }
}

int mi355_get_option(mi355_core *c, int option, int *value) {
    if (!c || !value) return fail(MI355_ERR_INVALID, "null argument");
    switch (option) {
        case MI355_OPT_PIPELINE: *value = c->pipeline_ok ? 1 : 0; return MI355_OK;
        case MI355_OPT_SPLIT_PCT: *value = c->split_pct; return MI355_OK;
        case MI355_OPT_DENSE_PCT: *value = c->dense_pct; return MI355_OK;
        case MI355_OPT_CHAIN_HINT: *value = c->chain_hint ? 1 : 0; return MI355_OK;
        case MI355_OPT_PACK_BLOCKS: *value = c->pack_blocks_opt; return MI355_OK;
        case MI355_OPT_MEDIAN_ROWS: *value = c->median_rows; return MI355_OK;
        case MI355_OPT_SCAN_EPOCH_LEFT: {
            const uint64_t left = kEpochWrap - 1 - c->scan_epoch;
            *value = left > (1ull << 30) ? (1 << 30) : (int)left;
            return MI355_OK;
        }
        default: return fail(MI355_ERR_INVALID, "unknown option");
    }
}

int mi355_set_state(mi355_core *c, const uint8_t *host_frame) {
    if (!c || !host_frame) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipMemcpyAsync(c->state, host_frame, c->n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MI355_OK;
}

int mi355_get_state(mi355_core *c, uint8_t *host_frame) {
    if (!c || !host_frame) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipMemcpyAsync(host_frame, c->state, c->n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MI355_OK;
}

void *mi355_state_device_ptr(mi355_core *c) { return c ? c->state : nullptr; }

int mi355_set_conv_kernel(mi355_core *c, const float *k9) {
    if (!c || !k9) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipMemcpyAsync(c->k9, k9, 9 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_k9 = true;
    uint32_t b[9];
    memcpy(b, k9, sizeof b);
    c->k9_sym = b[0] == b[2] && b[0] == b[6] && b[0] == b[8] && b[1] == b[3] && b[1] == b[5] && b[1] == b[7];
    return MI355_OK;
}

int mi355_set_glyphs(mi355_core *c, const uint8_t *chars_px, int nglyphs, int glyph_h, int glyph_w,
                     const char *charset) {
    if (!c || !chars_px || !charset) return fail(MI355_ERR_INVALID, "null argument");
    if (nglyphs < 0 || glyph_h < 0 || glyph_w < 0 || (int)strlen(charset) != nglyphs)
        return fail(MI355_ERR_INVALID, "glyph atlas / charset mismatch");
    if (int rc = use_device(c)) return rc;
    const size_t bytes = (size_t)nglyphs * 3 * glyph_h * glyph_w;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->glyphs) { (void)hipFree(c->glyphs); c->glyphs = nullptr; }
    HIP_TRY(hipMalloc((void **)&c->glyphs, bytes ? bytes : 16));
    HIP_TRY(hipMemcpy(c->glyphs, chars_px, bytes, hipMemcpyHostToDevice));
    c->nglyphs = nglyphs; c->glyph_h = glyph_h; c->glyph_w = glyph_w;
    c->charset = charset;
    return MI355_OK;
}

int mi355_diff_stream_batch(mi355_core *c, const void *d_frames, size_t stride_bytes, int nframes,
                            void *d_offsets, void *d_xs, void *d_diff, size_t capacity) {
    return run_batch(c, false, d_frames, nullptr, stride_bytes, nframes, d_offsets, d_xs, d_diff, capacity, nullptr, true);
}

int mi355_diff_pairs_batch(mi355_core *c, const void *d_cur, const void *d_prev, size_t stride_bytes,
                           int nframes, void *d_offsets, void *d_xs, void *d_diff, size_t capacity) {
    return run_batch(c, true, d_cur, d_prev, stride_bytes, nframes, d_offsets, d_xs, d_diff, capacity, nullptr, true);
}

int mi355_diff_stream_wire_batch(mi355_core *c, const void *d_frames, size_t stride_bytes, int nframes,
                                 void *d_offsets, void *d_wire, size_t capacity_bytes) {
    if (!d_wire) return fail(MI355_ERR_INVALID, "null d_wire");
    return run_batch(c, false, d_frames, nullptr, stride_bytes, nframes, d_offsets, nullptr, nullptr,
                     capacity_bytes, d_wire, true);
}

size_t mi355_wire_bytes(int nframes, uint64_t entries) { return 4 * (size_t)nframes + 5 * (size_t)entries; }

int mi355_apply_batch(mi355_core *c, const void *d_offsets, const void *d_xs, const void *d_diff, int nframes,
                      void *d_frames_out, size_t stride_bytes) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (nframes < 0) return fail(MI355_ERR_INVALID, "nframes < 0");
    if (nframes == 0 || c->n == 0) return MI355_OK;
    if (!d_offsets || !d_xs || !d_diff) return fail(MI355_ERR_INVALID, "null stream pointer");
    if (d_frames_out && stride_bytes < c->n) return fail(MI355_ERR_INVALID, "stride_bytes < frame bytes");
    if (int rc = use_device(c)) return rc;
    if (!d_frames_out) {
        HIP_TRY(launch_apply_all(c->state, c->n, (const int32_t *)d_xs, (const uint8_t *)d_diff,
                                 (const uint32_t *)d_offsets, nframes, c->stream));
        return MI355_OK;
    }
    for (int t = 0; t < nframes; t++) {
        HIP_TRY(launch_apply(c->state, c->n, d_xs, d_diff, (const uint32_t *)d_offsets, t, 0, c->stream));
        HIP_TRY(hipMemcpyAsync((uint8_t *)d_frames_out + (size_t)t * stride_bytes, c->state, c->n,
                               hipMemcpyDeviceToDevice, c->stream));
    }
    return MI355_OK;
}

int mi355_apply_wire_batch(mi355_core *c, const void *d_wire, const uint32_t *h_counts, int nframes,
                           void *d_frames_out, size_t stride_bytes) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (nframes < 0) return fail(MI355_ERR_INVALID, "nframes < 0");
    if (nframes == 0) return MI355_OK;
    if (!d_wire || !h_counts) return fail(MI355_ERR_INVALID, "null stream pointer");
    if (d_frames_out && stride_bytes < c->n) return fail(MI355_ERR_INVALID, "stride_bytes < frame bytes");
    if (int rc = use_device(c)) return rc;
    const uint8_t *p = (const uint8_t *)d_wire;
    for (int t = 0; t < nframes; t++) {
        const uint32_t cnt = h_counts[t];
        if (cnt > c->n) return fail(MI355_ERR_INVALID, "frame count larger than the frame");
        const uint8_t *xs = p + 4, *df = xs + 4 * (size_t)cnt;   // opencv.cpp:52-62
        HIP_TRY(launch_apply(c->state, c->n, xs, df, nullptr, 0, cnt, c->stream));
        if (d_frames_out && c->n)
            HIP_TRY(hipMemcpyAsync((uint8_t *)d_frames_out + (size_t)t * stride_bytes, c->state, c->n,
                                   hipMemcpyDeviceToDevice, c->stream));
        p = df + cnt;
    }
    return MI355_OK;
}

int mi355_merge_parts(mi355_core *c, int nparts, int nframes, const void *d_part_offsets,
                      const uint32_t *h_part_base, const int32_t *h_xs_bias, const void *d_xs_all,
                      const void *d_diff_all, void *d_offsets, void *d_xs, void *d_diff, size_t capacity) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (nparts < 1 || nparts > kMaxParts) return fail(MI355_ERR_INVALID, "nparts outside [1, 64]");
    if (nframes < 0) return fail(MI355_ERR_INVALID, "nframes < 0");
    if (!d_part_offsets || !h_part_base || !h_xs_bias || !d_offsets)
        return fail(MI355_ERR_INVALID, "null argument");
    if (capacity > 0 && (!d_xs_all || !d_diff_all || !d_xs || !d_diff))
        return fail(MI355_ERR_INVALID, "null stream pointer");
    if (int rc = use_device(c)) return rc;
    MergeArgs a{};
    a.part_off = (const uint32_t *)d_part_offsets;
    a.xs_all = (const int32_t *)d_xs_all;
    a.diff_all = (const uint8_t *)d_diff_all;
    a.out_xs = (int32_t *)d_xs;
    a.out_diff = (uint8_t *)d_diff;
    a.capacity = capacity;
    a.nparts = nparts;
    a.nframes = nframes;
    for (int p = 0; p < nparts; p++) {
        a.part_base[p] = h_part_base[p];
        a.xs_bias[p] = h_xs_bias[p];
    }
    HIP_TRY(launch_merge(a, (uint32_t *)d_offsets, c->stream));
    return MI355_OK;
}

int mi355_int_diff(mi355_core *c, const void *d_cur, const void *d_prev, void *d_out, size_t n) {
    if (!c || (n && (!d_cur || !d_prev || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device_filter(c)) return rc;
    HIP_TRY(launch_int_diff((const int32_t *)d_cur, (const int32_t *)d_prev, (int32_t *)d_out, n, c->stream));
    return MI355_OK;
}

int mi355_gray_avg(mi355_core *c, const void *d_in, void *d_out) {
    if (!c || (c->n && (!d_in || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device_filter(c)) return rc;
    HIP_TRY(launch_gray((const uint8_t *)d_in, (uint8_t *)d_out, c->n / 3, false, FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_gray_weighted(mi355_core *c, const void *d_in, void *d_out) {
    if (!c || (c->n && (!d_in || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device_filter(c)) return rc;
    HIP_TRY(launch_gray((const uint8_t *)d_in, (uint8_t *)d_out, c->n / 3, true, FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_binarize_chain(mi355_core *c, const void *d_gray, void *d_out, void *d_hist, void *d_thr) {
    if (!c || (c->n && (!d_gray || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device_filter(c)) return rc;
    int32_t *hist = d_hist ? (int32_t *)d_hist : c->hist;
    int32_t *thr = d_thr ? (int32_t *)d_thr : c->thr;
    HIP_TRY(launch_binarize_chain((const uint8_t *)d_gray, (uint8_t *)d_out, c->n, hist, thr, FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_heat_map(mi355_core *c, const void *d_cur, const void *d_prev, void *d_out) {
    if (!c || (c->n && (!d_cur || !d_prev || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device_filter(c)) return rc;
    HIP_TRY(launch_heat_map((const uint8_t *)d_cur, (const uint8_t *)d_prev, (uint8_t *)d_out, c->n / 3,
                            c->lut, FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_red_dense(mi355_core *c, const void *d_cur, const void *d_prev, void *d_out) {
    if (!c || (c->n && (!d_cur || !d_prev || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device_filter(c)) return rc;
    HIP_TRY(launch_red_dense((const uint8_t *)d_cur, (const uint8_t *)d_prev, (uint8_t *)d_out, c->n / 3,
                             c->cfg.threshold, FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_red_overlap(mi355_core *c, void *d_img, const void *d_xs, const void *d_count, uint32_t count) {
    if (!c || !d_img || ((d_count || count) && !d_xs)) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(launch_red_overlap((uint8_t *)d_img, (const int32_t *)d_xs, (const uint32_t *)d_count, count,
                               c->n, c->stream));
    return MI355_OK;
}

int mi355_red_stream_batch(mi355_core *c, const void *d_offsets, const void *d_xs, int nframes, void *d_frames,
                           size_t stride_bytes, int clear) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (nframes < 0) return fail(MI355_ERR_INVALID, "nframes < 0");
    if (nframes == 0 || c->n == 0) return MI355_OK;
    if (!d_offsets || !d_xs || !d_frames) return fail(MI355_ERR_INVALID, "null argument");
    if (stride_bytes < c->n) return fail(MI355_ERR_INVALID, "stride_bytes < frame bytes");
    if (int rc = use_device(c)) return rc;
    // only the cleared form needs per-frame scratch (slice bounds), which is sized for max_batch frames
    if (clear && nframes > c->cfg.max_batch) return fail(MI355_ERR_INVALID, "nframes outside [0, max_batch]");
    if (clear)
        if (int rc = need_red_bounds(c)) return rc;
    HIP_TRY(launch_red_stream((uint8_t *)d_frames, (const uint32_t *)d_offsets, (const int32_t *)d_xs, c->n, clear != 0,
                              FrameBatch{stride_bytes, nframes}, c->stream, c->red_bounds));
    return MI355_OK;
}

int mi355_conv3x3(mi355_core *c, const void *d_in, void *d_out) {
    if (!c || (c->n && (!d_in || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (d_in == d_out && c->n) return fail(MI355_ERR_INVALID, "conv3x3 cannot run in place");
    if (!c->have_k9) return fail(MI355_ERR_STATE, "mi355_set_conv_kernel not called");
    if (int rc = use_device_filter(c)) return rc;
    HIP_TRY(launch_conv3x3((const uint8_t *)d_in, (uint8_t *)d_out, c->cfg.width, c->cfg.height, c->k9,
                           c->k9_sym, FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_conv_kxk(mi355_core *c, const void *d_in, void *d_out, const float *k, int K) {
    if (!c || !k || (c->n && (!d_in || !d_out))) return fail(MI355_ERR_INVALID, "null argument");
    if (K < 1 || K > 9) return fail(MI355_ERR_INVALID, "K outside [1, 9]");
    if (d_in == d_out && c->n) return fail(MI355_ERR_INVALID, "conv_kxk cannot run in place");
    if (int rc = use_device_filter(c)) return rc;
    if (!c->kxk)
        if (int rc = dev_alloc(c, &c->kxk, 81)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));    // a filter of an earlier call may still be reading the taps
    HIP_TRY(hipMemcpyAsync(c->kxk, k, sizeof(float) * K * K, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));    // k is the caller's (pageable) memory
    HIP_TRY(launch_conv_kxk((const uint8_t *)d_in, (uint8_t *)d_out, c->cfg.width, c->cfg.height, c->kxk, K,
                            FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_median5x5(mi355_core *c, const void *d_in, void *d_out) {
    if (!c || !d_in || !d_out) return fail(MI355_ERR_INVALID, "null argument");
    if (d_in == d_out) return fail(MI355_ERR_INVALID, "median is not in-place");
    if (int rc = use_device_filter(c)) return rc;
    HIP_TRY(launch_median5x5((const uint8_t *)d_in, (uint8_t *)d_out, c->cfg.width, c->cfg.height, c->median_rows,
                             FrameBatch{c->n, 1}, c->stream));
    return MI355_OK;
}

int mi355_filter_batch(mi355_core *c, int op, const void *d_in, const void *d_in2, void *d_out,
                       size_t stride_bytes, int nframes) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (nframes < 0 || nframes > c->cfg.max_batch) return fail(MI355_ERR_INVALID, "nframes outside [0, max_batch]");
    if (nframes == 0 || c->n == 0) return MI355_OK;
    if (!d_in || !d_out) return fail(MI355_ERR_INVALID, "null frame pointer");
    if (stride_bytes < c->n) return fail(MI355_ERR_INVALID, "stride_bytes < frame bytes");
    const bool two = op == MI355_OP_HEAT_MAP || op == MI355_OP_RED_DENSE;
    if (two && !d_in2) return fail(MI355_ERR_INVALID, "this filter needs the previous frames (d_in2)");
    if (op == MI355_OP_CONV3X3 && !c->have_k9) return fail(MI355_ERR_STATE, "mi355_set_conv_kernel not called");
    if ((op == MI355_OP_CONV3X3 || op == MI355_OP_MEDIAN5X5) && d_in == d_out)
        return fail(MI355_ERR_INVALID, "neighbourhood filters cannot run in place");
    if (int rc = use_device_filter(c)) return rc;
    const uint8_t *in = (const uint8_t *)d_in, *in2 = (const uint8_t *)d_in2;
    uint8_t *out = (uint8_t *)d_out;
    const FrameBatch fb{stride_bytes, nframes};
    const uint32_t npix = c->n / 3;
    switch (op) {
        case MI355_OP_GRAY_AVG: HIP_TRY(launch_gray(in, out, npix, false, fb, c->stream)); break;
        case MI355_OP_GRAY_WEIGHTED: HIP_TRY(launch_gray(in, out, npix, true, fb, c->stream)); break;
        case MI355_OP_BINARIZE: HIP_TRY(launch_binarize_chain(in, out, c->n, c->hist, c->thr, fb, c->stream)); break;
        case MI355_OP_GRAY_AVG_BINARIZE:
            if (int rc = need_gray1(c)) return rc;
            HIP_TRY(launch_gray_binarize_fused(in, out, npix, false, c->hist, c->thr, fb, c->stream, c->gray1, c->gray1_stride)); break;
        case MI355_OP_GRAY_WEIGHTED_BINARIZE:
            if (int rc = need_gray1(c)) return rc;
            HIP_TRY(launch_gray_binarize_fused(in, out, npix, true, c->hist, c->thr, fb, c->stream, c->gray1, c->gray1_stride)); break;
        case MI355_OP_HEAT_MAP: HIP_TRY(launch_heat_map(in, in2, out, npix, c->lut, fb, c->stream)); break;
        case MI355_OP_RED_DENSE: HIP_TRY(launch_red_dense(in, in2, out, npix, c->cfg.threshold, fb, c->stream)); break;
        case MI355_OP_CONV3X3: HIP_TRY(launch_conv3x3(in, out, c->cfg.width, c->cfg.height, c->k9, c->k9_sym, fb, c->stream)); break;
        case MI355_OP_MEDIAN5X5: HIP_TRY(launch_median5x5(in, out, c->cfg.width, c->cfg.height, c->median_rows, fb, c->stream)); break;
        default: return fail(MI355_ERR_INVALID, "unknown filter op");
    }
    return MI355_OK;
}

// CUDACore::exec_core, kernels.cu:430-525.
namespace {

// kernels.cu:466-502: text overlay on the frame, then the visualisers that look at the frame before the
// diff (heat / gray / binarize), or the canvas the red maps are painted on afterwards.
int prepare_frame(mi355_core *c, uint8_t *frame, uint8_t *vis_out, const char *text, hipStream_t s) {
    const int vis = c->cfg.visualizer;
    const uint32_t N = c->n, npix = N / 3;
    const FrameBatch one{N, 1};
    if (text && c->glyphs) {
        const size_t full_area = (size_t)3 * c->glyph_h * c->glyph_w;
        int offset = 0;
        for (const char *p = text; *p; ++p, offset += c->glyph_w * 3) {
            const size_t idx = c->charset.find(*p);
            if (idx == std::string::npos) continue;
            HIP_TRY(launch_blit_glyph(frame, c->glyphs + idx * full_area, c->glyph_h, 3 * c->glyph_w, offset,
                                      3 * c->cfg.width, c->cfg.height, s));
        }
    }
    if (vis == MI355_VIS_HEAT) {
        HIP_TRY(launch_heat_map(frame, c->state, vis_out, npix, c->lut, one, s));
    } else if (vis == MI355_VIS_GRAY) {
        HIP_TRY(launch_gray(frame, vis_out, npix, true, one, s));
    } else if (vis == MI355_VIS_BINARIZE) {
        // grayscale_kernel_v3 + histogram + compute_max + binarize (kernels.cu:493-498), fused: the
        // gray frame is never materialised
        if (int rc = need_gray1(c)) return rc;
        HIP_TRY(launch_gray_binarize_fused(frame, vis_out, npix, true, c->hist, c->thr, one, s, c->gray1, c->gray1_stride));
    } else if (vis == MI355_VIS_RED_OVERLAP) {
        // kernels.cu:517 paints onto d_previous, i.e. the state *before* this frame's feedback
        HIP_TRY(hipMemcpyAsync(vis_out, c->state, N, hipMemcpyDeviceToDevice, s));
    } else if (vis == MI355_VIS_RED) {
        HIP_TRY(hipMemsetAsync(vis_out, 0, N, s));                      // kernels.cu:513
    }
    return MI355_OK;
}

int check_exec_args(mi355_core *c, const void *frame_data, const void *show_ready, const void *h_xs) {
    if (!c || !frame_data || !h_xs) return fail(MI355_ERR_INVALID, "null argument");
    if (c->cfg.visualizer != MI355_VIS_NONE && !show_ready)
        return fail(MI355_ERR_INVALID, "visualizer set but show_ready is null");
    if (c->cfg.noise_filter && !c->have_k9) return fail(MI355_ERR_STATE, "noise filter on but no conv kernel set");
    return MI355_OK;
}

bool is_pinned(const void *p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

}  // namespace

// MI355_PREPARE_EXEC: the kernels of one mi355_exec over a copy of the current state.  Nothing differs from the state, so no
// byte is flagged, the state is rewritten with itself and the export stores a count of 0 into the core's own pinned word.
static int warm_exec(mi355_core *c) {
    hipStream_t s = c->stream;
    const uint32_t N = c->n;
    const int vis = c->cfg.visualizer;
    HIP_TRY(hipMemcpyAsync(c->in, c->state, N, hipMemcpyDeviceToDevice, s));
    if (c->cfg.noise_filter && c->have_k9)   // (its output is not used: a filtered frame would differ from the state)
        HIP_TRY(launch_conv3x3(c->in, c->aux, c->cfg.width, c->cfg.height, c->k9, c->k9_sym, FrameBatch{N, 1}, s));
    if (int rc = prepare_frame(c, c->in, c->vis, nullptr, s)) return rc;
    if (int rc = run_batch(c, false, c->in, nullptr, N, 1, c->offsets, c->one_xs, c->one_diff, N)) return rc;
    if (vis == MI355_VIS_RED || vis == MI355_VIS_RED_OVERLAP)
        HIP_TRY(launch_red_overlap(c->vis, c->one_xs, c->offsets + 1, 0, N, s));
    HIP_TRY(launch_export(c->offsets, c->one_xs, c->one_diff, (int32_t *)c->h_count, (uint8_t *)c->h_count, c->h_count, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (c->h_count[0] != 0) return fail(MI355_ERR_STATE, "mi355_prepare(EXEC): the warm pass found a difference");
    return MI355_OK;
}

int mi355_exec(mi355_core *c, uint8_t *frame_data, uint8_t *show_ready, const char *text,
               uint32_t *h_pos, int32_t *h_xs) {
    if (!h_pos) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = check_exec_args(c, frame_data, show_ready, h_xs)) return rc;
    if (c->nslots) return fail(MI355_ERR_STATE, "pipe open: use mi355_pipe_submit");
    const int vis = c->cfg.visualizer;
    if (int rc = use_device(c)) return rc;
    hipStream_t s = c->stream;
    const uint32_t N = c->n;
    const FrameBatch one{N, 1};

    // kernels.cu:457-462  H2D (+ convolution when NOISE_FILTER)
    if (c->cfg.noise_filter) {
        HIP_TRY(hipMemcpyAsync(c->aux, frame_data, N, hipMemcpyHostToDevice, s));
        HIP_TRY(launch_conv3x3(c->aux, c->in, c->cfg.width, c->cfg.height, c->k9, c->k9_sym, one, s));
    } else {
        HIP_TRY(hipMemcpyAsync(c->in, frame_data, N, hipMemcpyHostToDevice, s));
    }
    if (int rc = prepare_frame(c, c->in, c->vis, text, s)) return rc;
    if (vis == MI355_VIS_HEAT || vis == MI355_VIS_GRAY || vis == MI355_VIS_BINARIZE)
        HIP_TRY(hipMemcpyAsync(show_ready, c->vis, N, hipMemcpyDeviceToHost, s));

    // kernels.cu:505  kernel2
    if (int rc = run_batch(c, false, c->in, nullptr, N, 1, c->offsets, c->one_xs, c->one_diff, N)) return rc;
    if (is_pinned(frame_data) && is_pinned(h_xs)) {
        // Pinned buffers (alloc_arrays): the count stays on the device, the red maps take it from there and
        // k_export stores count/diff/xs through the mapped pointers -- one synchronisation instead of the
        // reference's two (kernels.cu:508,524); what the caller sees on return is the same.
        if (vis == MI355_VIS_RED || vis == MI355_VIS_RED_OVERLAP) {
            HIP_TRY(launch_red_overlap(c->vis, c->one_xs, c->offsets + 1, 0, N, s));
            HIP_TRY(hipMemcpyAsync(show_ready, c->vis, N, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(launch_export(c->offsets, c->one_xs, c->one_diff, h_xs, frame_data, c->h_count, s));
        HIP_TRY(hipStreamSynchronize(s));
        *h_pos = c->h_count[0];
        return MI355_OK;
    }
    // pageable buffers: the reference's sequence, kernels.cu:507-508 count back + first synchronisation
    HIP_TRY(hipMemcpyAsync(c->h_count, c->offsets, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t pos = c->h_count[1];
    *h_pos = pos;
    // kernels.cu:511-520  red maps from the packed indices
    if (vis == MI355_VIS_RED || vis == MI355_VIS_RED_OVERLAP) {
        HIP_TRY(launch_red_overlap(c->vis, c->one_xs, nullptr, pos, N, s));
        HIP_TRY(hipMemcpyAsync(show_ready, c->vis, N, hipMemcpyDeviceToHost, s));
    }
    // kernels.cu:522-524
    if (pos) {
        HIP_TRY(hipMemcpyAsync(frame_data, c->one_diff, pos, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h_xs, c->one_xs, (size_t)pos * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return MI355_OK;
}

// ---- pipelined per-frame path --------------------------------------------------------------------------
int mi355_pipe_close(mi355_core *c) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->nslots; i++) {
        mi355_core::Slot &sl = c->slots[i];
        if (sl.d_in) { (void)hipFree(sl.d_in); c->workspace -= c->n + 16; }
        if (sl.d_vis) { (void)hipFree(sl.d_vis); c->workspace -= c->n + 16; }
        if (sl.h_count) (void)hipHostFree(sl.h_count);
        for (hipEvent_t e : {sl.uploaded, sl.painted, sl.packed, sl.shown}) if (e) (void)hipEventDestroy(e);
        sl = mi355_core::Slot{};
    }
    c->nslots = 0;
    if (c->up_stream) { (void)hipStreamDestroy(c->up_stream); c->up_stream = nullptr; }
    if (c->down_stream) { (void)hipStreamDestroy(c->down_stream); c->down_stream = nullptr; }
    return MI355_OK;
}

int mi355_pipe_open(mi355_core *c, int depth) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (depth < 1 || depth > mi355_core::kMaxSlots) return fail(MI355_ERR_INVALID, "depth outside [1, 8]");
    if (c->nslots) return fail(MI355_ERR_STATE, "pipe already open");
    if (int rc = use_device(c)) return rc;
    int rc = MI355_OK;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking)) != hipSuccess) rc = fail(MI355_ERR_HIP, "hipStreamCreate", e);
    if (!rc && (e = hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking)) != hipSuccess) rc = fail(MI355_ERR_HIP, "hipStreamCreate", e);
    c->nslots = depth;   // so that a failed open is undone by pipe_close
    const bool vis = c->cfg.visualizer != MI355_VIS_NONE;
    for (int i = 0; i < depth && !rc; i++) {
        mi355_core::Slot &sl = c->slots[i];
        rc = dev_alloc(c, &sl.d_in, (size_t)c->n + 16);
        if (!rc && vis) rc = dev_alloc(c, &sl.d_vis, (size_t)c->n + 16);
        if (!rc && (e = hipHostMalloc((void **)&sl.h_count, sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) rc = fail(MI355_ERR_HIP, "hipHostMalloc", e);
        for (hipEvent_t *ev : {&sl.uploaded, &sl.painted, &sl.packed, &sl.shown})
            if (!rc && (e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) rc = fail(MI355_ERR_HIP, "hipEventCreate", e);
    }
    if (rc) {
        const std::string keep = g_err;
        (void)mi355_pipe_close(c);
        g_err = keep;
        return rc;
    }
    c->next_ticket = 0;
    return MI355_OK;
}

int mi355_pipe_submit(mi355_core *c, uint8_t *frame_data, uint8_t *show_ready, const char *text, int32_t *h_xs,
                      int64_t *ticket) {
    if (!ticket) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = check_exec_args(c, frame_data, show_ready, h_xs)) return rc;
    if (!c->nslots) return fail(MI355_ERR_STATE, "pipe not open");
    if (int rc = use_device(c)) return rc;
    // the kernels store through these pointers: pageable memory would fault on the device
    if (!is_pinned(frame_data) || !is_pinned(h_xs) || (show_ready && !is_pinned(show_ready)))
        return fail(MI355_ERR_INVALID, "pipe buffers must be pinned host memory (mi355_host_alloc)");
    const int vis = c->cfg.visualizer;
    const uint32_t N = c->n;
    const FrameBatch one{N, 1};
    mi355_core::Slot &sl = c->slots[c->next_ticket % c->nslots];
    hipStream_t s = c->stream;
    if (sl.ticket >= 0) {   // ring full: the slot's previous frame was never waited for; finish it first
        HIP_TRY(hipEventSynchronize(sl.packed));
        if (sl.has_vis) HIP_TRY(hipEventSynchronize(sl.shown));
        sl.ticket = -1;
    }
    // upload on its own stream: frame k+1 crosses PCIe while frame k is packed (threads.cpp's capture and
    // elaboration threads overlap the same way, threads.cpp:59-106,134-147)
    HIP_TRY(hipMemcpyAsync(sl.d_in, frame_data, N, hipMemcpyHostToDevice, c->up_stream));
    HIP_TRY(hipEventRecord(sl.uploaded, c->up_stream));
    HIP_TRY(hipStreamWaitEvent(s, sl.uploaded, 0));
    uint8_t *frame = sl.d_in;
    if (c->cfg.noise_filter) {
        HIP_TRY(launch_conv3x3(sl.d_in, c->in, c->cfg.width, c->cfg.height, c->k9, c->k9_sym, one, s));
        frame = c->in;
    }
    if (int rc = prepare_frame(c, frame, sl.d_vis, text, s)) return rc;
    if (int rc = run_batch(c, false, frame, nullptr, N, 1, c->offsets, c->one_xs, c->one_diff, N)) return rc;
    if (vis == MI355_VIS_RED || vis == MI355_VIS_RED_OVERLAP)
        HIP_TRY(launch_red_overlap(sl.d_vis, c->one_xs, c->offsets + 1, 0, N, s));
    sl.has_vis = vis != MI355_VIS_NONE;
    if (sl.has_vis) {   // the visualisation frame goes back on a third stream, beside the next frame's kernels
        HIP_TRY(hipEventRecord(sl.painted, s));
        HIP_TRY(hipStreamWaitEvent(c->down_stream, sl.painted, 0));
        HIP_TRY(hipMemcpyAsync(show_ready, sl.d_vis, N, hipMemcpyDeviceToHost, c->down_stream));
        HIP_TRY(hipEventRecord(sl.shown, c->down_stream));
    }
    // count, indices and differences leave through the mapped pointers: no host round trip for the count
    HIP_TRY(launch_export(c->offsets, c->one_xs, c->one_diff, h_xs, frame_data, sl.h_count, s));
    HIP_TRY(hipEventRecord(sl.packed, s));
    sl.ticket = c->next_ticket;
    *ticket = c->next_ticket++;
    return MI355_OK;
}

int mi355_pipe_wait(mi355_core *c, int64_t ticket, uint32_t *h_pos) {
    if (!c || !h_pos) return fail(MI355_ERR_INVALID, "null argument");
    if (!c->nslots) return fail(MI355_ERR_STATE, "pipe not open");
    if (ticket < 0 || ticket >= c->next_ticket) return fail(MI355_ERR_INVALID, "unknown ticket");
    mi355_core::Slot &sl = c->slots[ticket % c->nslots];
    if (sl.ticket != ticket) return fail(MI355_ERR_STATE, "ticket already waited for or overwritten");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipEventSynchronize(sl.packed));
    if (sl.has_vis) HIP_TRY(hipEventSynchronize(sl.shown));
    *h_pos = *sl.h_count;
    sl.ticket = -1;
    return MI355_OK;
}

int mi355_host_alloc(void **out, size_t bytes) {
    if (!out) return fail(MI355_ERR_INVALID, "null argument");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return MI355_OK;
}

int mi355_host_free(void *p) {
    if (!p) return MI355_OK;
    HIP_TRY(hipHostFree(p));
    return MI355_OK;
}

int mi355_dev_alloc(mi355_core *c, void **out, size_t bytes) {
    if (!c || !out) return fail(MI355_ERR_INVALID, "null argument");
    *out = nullptr;
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
    return MI355_OK;
}

// ---- output arrays whose PLACEMENT lets the dense expansion run at its fast speed ---------------------------------------
// What round 6 found (profiles/README.md, r06a-r06e): when most bytes of a frame change (the synthetic S0 / P = N regimes, a
// scene cut), k_expand is bound by its stores, and writes the index array and the value array at 5.1 TB/s together when the
// two streams overlap in the memory system -- or at 4.0 TB/s (265 instead of 205 us per 32 S0 pairs) when they do not.
// Which of the two a pair of arrays gets is a property of the PAIR's physical memory (it stays with the arrays for their
// life; another value array beside the same index array re-draws it; between one pair in six and two in three are of the
// fast kind, box to box; physically contiguous memory is the slowest; the L2's tag stalls and the DRAM-credit stalls of
// its write requests differ, the request counts do not) -- nothing a caller can see in an address, nothing an offset
// inside an allocation changes, and a synthetic store kernel of another grid shape mis-predicts it (r06d).  What a caller
// CAN do is measure with the expansion itself: this call packs a batch of noise frames (P = 0.85 N, the S0 regime) once per
// candidate value array and keeps the first pair on which k_expand moves its bytes at the fast rate.
namespace {
__global__ __launch_bounds__(256) void k_noise_frames(uint32_t *out, size_t nwords, uint32_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        out[i] = x;
    }
}
}  // namespace

int mi355_alloc_outputs(mi355_core *c, size_t capacity, void **d_xs, void **d_diff, int *draws_out) {
    if (!c || !d_xs || !d_diff) return fail(MI355_ERR_INVALID, "null argument");
    *d_xs = *d_diff = nullptr;
    if (draws_out) *draws_out = 0;
    if (c->nslots) return fail(MI355_ERR_STATE, "pipe open");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t xs_bytes = (capacity ? capacity : 4) * sizeof(int32_t), df_bytes = capacity ? capacity : 16;
    int32_t *xs = nullptr;
    HIP_TRY(hipMalloc((void **)&xs, xs_bytes));
    // the probe: T frames of noise against T other frames of noise, as many as the arrays hold; only worth its while where the
    // placement matters (a pair of a few hundred MB written by a few thousand expander waves at a time)
    const size_t N = c->n;
    int T = c->cfg.max_batch;
    if (N && (size_t)T * N > capacity) T = (int)(capacity / N);
    constexpr int kMaxDraws = 32;
    uint8_t *cand[kMaxDraws] = {};
    int ndraw = 0, keep = -1;
    int rc = MI355_OK;
    if (T >= 8 && (size_t)T * N >= ((size_t)48 << 20) && N % 16 == 0 && 3 * (uint64_t)N + N < (1ull << 32)) {
        uint8_t *frames = nullptr;
        uint32_t *off = nullptr;
        const size_t fbytes = 2 * (size_t)T * N;
        if (hipMalloc((void **)&frames, fbytes) != hipSuccess || hipMalloc((void **)&off, ((size_t)T + 1) * 4) != hipSuccess) {
            (void)hipGetLastError();   // no room for the probe: plain allocations
        } else {
            hipLaunchKernelGGL(k_noise_frames, dim3(4096), dim3(256), 0, c->stream, (uint32_t *)frames, fbytes / 4, 0x9e3779b9u);
            const bool was_timing = c->timing;
            if ((rc = harvest_timing(c)) == MI355_OK) {
                const double s_pack = c->ms_pack, s_scan = c->ms_scan, s_exp = c->ms_expand, s_tot = c->ms_total;
                const int s_l = c->launches;
                double best_rate = 0;
                while (ndraw < kMaxDraws && rc == MI355_OK) {
                    if (hipMalloc((void **)&cand[ndraw], df_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
                    uint8_t *df = cand[ndraw++];
                    // untimed first: ~15 ms of this load on the first candidate (a chip that has been idle needs them to reach
                    // its clocks), one batch on the others
                    c->timing = false;
                    for (int rep = 0; rep < (ndraw == 1 ? 48 : 1) && rc == MI355_OK; rep++)
                        rc = run_batch(c, true, frames, frames + (size_t)T * N, N, T, off, xs, df, capacity);
                    c->timing = true;
                    c->ms_expand = 0; c->launches = 0;
                    for (int rep = 0; rep < 3 && rc == MI355_OK; rep++)
                        rc = run_batch(c, true, frames, frames + (size_t)T * N, N, T, off, xs, df, capacity);
                    if (rc == MI355_OK) rc = harvest_timing(c);
                    if (rc != MI355_OK) break;
                    uint32_t total = 0;
                    if (hipMemcpy(&total, off + T, 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MI355_ERR_HIP, "hipMemcpy(total)"); break; }
                    // bytes the expansion moves: the records it reads (16 per lane: T x N) + 5 per entry it writes
                    const double ms = c->ms_expand / (c->launches ? c->launches : 1);
                    const double rate = ((double)T * N + 5.0 * total) / (ms * 1e-3);   // bytes per second
                    if (rate > best_rate) { best_rate = rate; keep = ndraw - 1; }
                    if (rate >= 4.5e12) break;   // fast pairs: 5.0-5.2 TB/s; slow ones: 3.9-4.1 (profiles/r06*)
                }
                c->timing = was_timing;
                c->ms_pack = s_pack; c->ms_scan = s_scan; c->ms_expand = s_exp; c->ms_total = s_tot; c->launches = s_l;
            }
            (void)hipStreamSynchronize(c->stream);
        }
        if (frames) (void)hipFree(frames);
        if (off) (void)hipFree(off);
    }
    if (rc == MI355_OK && keep < 0) {   // small arrays, or no room for the probe
        if (ndraw == 0 && hipMalloc((void **)&cand[0], df_bytes) == hipSuccess) ndraw = 1;
        keep = ndraw > 0 ? 0 : -1;
        if (keep < 0) rc = fail(MI355_ERR_HIP, "hipMalloc(value array)");
    }
    for (int i = 0; i < ndraw; i++)
        if (i != keep || rc != MI355_OK) (void)hipFree(cand[i]);
    if (rc != MI355_OK) { (void)hipFree(xs); return rc; }
    *d_xs = xs;
    *d_diff = cand[keep];
    if (draws_out) *draws_out = ndraw;
    return MI355_OK;
}

int mi355_dev_free(mi355_core *c, void *d_ptr) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));   // nothing of this core may still be using it
    HIP_TRY(hipFree(d_ptr));
    return MI355_OK;
}

int mi355_upload(mi355_core *c, void *d_dst, const void *host_src, size_t bytes) {
    if (!c || (bytes && (!d_dst || !host_src))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device(c)) return rc;
    if (!bytes) return MI355_OK;
    HIP_TRY(hipMemcpyAsync(d_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MI355_OK;
}

int mi355_download(mi355_core *c, void *host_dst, const void *d_src, size_t bytes) {
    if (!c || (bytes && (!host_dst || !d_src))) return fail(MI355_ERR_INVALID, "null argument");
    if (int rc = use_device(c)) return rc;
    if (!bytes) return MI355_OK;
    HIP_TRY(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MI355_OK;
}

int mi355_set_timing(mi355_core *c, int enabled) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    if (int rc = harvest_timing(c)) return rc;
    c->timing = enabled != 0;
    return MI355_OK;
}

int mi355_get_timing(mi355_core *c, double *ms_pack, double *ms_total, int *launches) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    if (int rc = harvest_timing(c)) return rc;
    if (ms_pack) *ms_pack = c->ms_pack;
    if (ms_total) *ms_total = c->ms_total;
    if (launches) *launches = c->launches;
    return MI355_OK;
}

int mi355_get_kernel_timing(mi355_core *c, double *ms_pack, double *ms_scan, double *ms_expand, int *launches) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    if (int rc = harvest_timing(c)) return rc;
    if (ms_pack) *ms_pack = c->ms_pack;
    if (ms_scan) *ms_scan = c->ms_scan;
    if (ms_expand) *ms_expand = c->ms_expand;
    if (launches) *launches = c->launches;
    return MI355_OK;
}

int mi355_reset_timing(mi355_core *c) {
    if (!c) return fail(MI355_ERR_INVALID, "null core");
    if (int rc = use_device(c)) return rc;
    if (int rc = harvest_timing(c)) return rc;
    c->ms_pack = c->ms_scan = c->ms_expand = c->ms_total = 0;
    c->launches = 0;
    return MI355_OK;
}

}  // extern "C"
