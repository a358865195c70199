// diff_fused.hip -- EXPERIMENT (opt-in, MI355_FLAG_FUSED): the stream form of diff + threshold + feedback +
// ordered pack as ONE resident kernel.  Bit-exact with the log path (same tests), slower on the MI355X
// (0.96 ms against 0.62 ms per 256-frame 1080p batch, profiles/README.md "r02 fused kernel"): kept as the
// measured answer to "can the record log's round trip through HBM be removed", not as the product path.
//
// Same semantics as k_diff_pack + k_scan_groups + k_expand (diff_pack.hip; reference kernel2,
// server/src/kernels.cu:289-334, CPU statement tests/cuda_streaming/test.cu:560-576), without the record
// log (0.5 GB out and 0.6 GB back per 256-frame 1080p batch):
//
//   * a wave64 owns one 1 KiB tile for the whole batch, the tile's state in 4 VGPRs (as k_diff_pack);
//   * the records of a frame (16 masked diff bytes of each lane with a flagged byte) are appended to a
//     wave-private ring in LDS; whenever 64 are waiting they are turned, with all 64 lanes busy, into 4-byte
//     entries {byte index in the tile, frame in the epoch, difference} in a wave-private LDS FIFO;
//   * frames are processed in *epochs* of kEpoch frames.  At the end of an epoch a workgroup (4 tiles)
//     publishes its per-frame byte counts: wgsum[t][wg] = count | launch tag << 16, and an atomic add of
//     (count | 1 << 20) into gsum[t][group of 64 workgroups] -- both words say by themselves whether they are
//     complete (the tag; the number of contributors) -- and arrives at a two-level tree (64 workgroups per
//     group counter, one counter per epoch) whose last arriver raises a flag word for every workgroup, each
//     in a 128-byte line of its own.  One epoch later -- after it has packed the NEXT epoch -- a workgroup
//     waits for its flag, reads the row of gsum and its group's 64 words of wgsum for every frame of the
//     epoch, derives its own prefix per frame and writes its entries to their final place in (xs, diff): the
//     output never exists in any other form in HBM.
//   * a workgroup only ever polls its OWN flag line.  A word polled by a thousand waves (an arrival counter,
//     a shared row, even a status word) starves its memory channel for everybody: measured, 2.8-3.1 ms per
//     batch with any of those instead of 0.96 ms.
//   * a frame in which a tile has more than kRawRecords candidate lanes (a moving object's body) bypasses
//     LDS: the 64 records go to a per-tile spill slot as they stand (one coalesced 1 KiB store) and are
//     expanded from there at write-out; FIFO entries beyond its LDS capacity overflow to HBM the same way.
//
// Why it loses (ablations in profiles/r02a_fused_ablation.log): compare + counts alone 0.32 ms; + staging in
// LDS 0.38; + expansion rounds 0.48 (the kernel is VALU-bound: work moved into it is not free); + output
// stores 0.59; + the exchange 0.96: publishing, the tree, the flag and the rows are ~6 dependent agent-scope
// round trips (~20 us) per epoch of ~10 us, and with 6 workgroups per CU nothing hides them.  Hiding them
// needs 3+ epochs of entries in LDS (it has room for 2 at this occupancy) and would still leave 0.59 ms.
//
// Every workgroup of the grid must be resident at the same time (flags are waited for):
// launch_diff_fused's caller checks the grid against the occupancy of the device (fused_capacity), sizes
// that do not fit use the log path.  A wait is bounded (kSpinLimitTicks): on expiry the kernel raises
// status[0] and every workgroup leaves, so the grid always drains.
#include "pack_common.h"

namespace mi355 {

#ifndef MI355_FUSED_ABLATE
#define MI355_FUSED_ABLATE 0   // timing builds only: 1 = no cross-workgroup exchange (prefix 0), 2 = also no output stores,
                               // 3 = also no expansion rounds, 4 = also no record staging (compare + counts only)
#endif

constexpr int kEpoch = 8;                 // frames per epoch = two register groups
constexpr int kGroupFrames = 2;           // frames per register group (two groups in flight)
constexpr uint32_t kRing = 128;           // staged records per wave (power of two; < 64 wait + <= kRawRecords arrive)
constexpr uint32_t kFifo = 384;           // entries per wave and epoch kept in LDS
constexpr uint32_t kRawRecords = 32;      // more candidate lanes than this in one frame of a tile: raw spill
constexpr uint32_t kOvf = kEpoch * kRawRecords * 16;   // entries the staged frames of one epoch can produce at most
constexpr uint32_t kWgPerGroup = 64;      // workgroups per gsum group
constexpr uint64_t kSpinLimitTicks = 200000000ull;     // 2 s of the 100 MHz wall clock
static_assert(kEpoch % (2 * kGroupFrames) == 0, "an epoch is a whole number of register-group pairs");
static_assert(kRawRecords + 63 < kRing, "ring must hold a partial round plus one frame");
static_assert(kEpoch == 2 * (int)kWavesPerBlock, "write-out: every wave derives the prefix of two frames of the epoch");

struct WaveLds {
    uint4 rec[kRing];            // staged records
    uint32_t fifo[2][kFifo];     // entries of the epoch being packed / the epoch being written out
    uint16_t rsrc[kRing];        // lane | frame-in-epoch << 6 of each staged record
    uint32_t n[2][kEpoch];       // flagged bytes of this tile per frame of the epoch
};

struct WaveBook {   // wave-uniform bookkeeping
    uint32_t head = 0, tail = 0;   // staging ring (free running)
    uint32_t ftail = 0;            // entries in the FIFO of the epoch being packed
    uint32_t rawmask = 0;          // frames of that epoch spilled raw
};

// LDS-only workgroup barrier: __syncthreads() would also wait for the frame loads in flight (its fence
// covers global memory); the hand-offs here are all through LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Cross-lane hand-off through LDS inside one wave: DS operations of a wave execute in order, this only
// keeps the compiler from moving accesses across the hand-off.
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t agent_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane(wave_inclusive_scan((int)v), 63);
}

__device__ __forceinline__ void fifo_put(const FusedArgs &a, WaveLds &w, uint32_t eb, uint32_t tile, uint32_t p, uint32_t v) {
    if (p < kFifo) w.fifo[eb][p] = v;
    else a.ovf[((size_t)tile * 2 + eb) * kOvf + (p - kFifo)] = v;
}

// Up to 64 staged records -> entries, every lane one record (the form k_expand uses, fed from LDS).
__device__ __forceinline__ void expand_round(const FusedArgs &a, WaveLds &w, WaveBook &b, uint32_t eb,
                                             uint32_t tile, uint32_t nvalid, uint32_t lane) {
    wave_lds_handoff();
    const uint32_t slot = (b.head + lane) & (kRing - 1);
    uint4 rec = w.rec[slot];
    const uint32_t src = w.rsrc[slot];
    if (lane >= nvalid) rec = make_uint4(0, 0, 0, 0);
    uint32_t m16 = record_map16(rec);
    const uint32_t cnt = (uint32_t)__builtin_popcount(m16);
    const uint32_t base = ((src & 63u) << 4) | ((src >> 6) << 10);
    if (__ballot(cnt > 1u) == 0) {
        // the usual round: isolated bytes, one per record -- no scan, no walk; the byte is the sum of the
        // record's bytes
        const uint32_t p = b.ftail + lane;
        const uint32_t byte = __builtin_amdgcn_sad_u8((rec.x | rec.y) | (rec.z | rec.w), 0u, 0u);
        const uint32_t v = (base + (uint32_t)__builtin_ctz(m16 | 0x10000u)) | (byte << 16);
        if (lane < nvalid) {
            if (b.ftail + nvalid <= kFifo) w.fifo[eb][p] = v;
            else fifo_put(a, w, eb, tile, p, v);
        }
        b.ftail += nvalid;
    } else {
        const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t p = b.ftail + incl - cnt;
        const uint8_t *bytes = reinterpret_cast<const uint8_t *>(&w.rec[slot]);
        const bool fits = b.ftail + total <= kFifo;    // wave-uniform
        while (m16) {
            const int bit = __builtin_ctz(m16);
            m16 &= m16 - 1;
            const uint32_t v = (base + (uint32_t)bit) | ((uint32_t)bytes[bit] << 16);
            if (fits) w.fifo[eb][p] = v;
            else fifo_put(a, w, eb, tile, p, v);
            ++p;
        }
        b.ftail += total;
    }
    b.head += nvalid;
    wave_lds_handoff();
}

// One frame of one tile.  Returns through cnt4: 4 * (this lane's flagged bytes) + 24 (see pack_step).
__device__ __forceinline__ void fused_step(const FusedArgs &a, const uint4 c, uint4 &s, ThrConst tc, WaveLds &w,
                                           WaveBook &b, uint32_t eb, uint32_t f, uint32_t tile, uint32_t lane,
                                           uint32_t &cnt4) {
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
    uint32_t sw[4] = {s.x, s.y, s.z, s.w};
    uint32_t dm[4], sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t x;
        const uint32_t fh = dword_flags(cw[k], sw[k], tc, x);
        sel[k] = perm_select(fh);
        const uint32_t d = bytes_sub_from_x(cw[k], sw[k], x);
        dm[k] = __builtin_amdgcn_perm(d, 0u, sel[k]);            // kernels.cu:314, 0 where un-flagged
        sw[k] = __builtin_amdgcn_perm(cw[k], sw[k], sel[k]);     // negative feedback, kernels.cu:316-331
    }
    s = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    cnt4 = __builtin_amdgcn_sad_u8(sel[0] + sel[1] + sel[2] + sel[3], 0u, 0u);

    const bool cand = ((dm[0] | dm[1]) | (dm[2] | dm[3])) != 0u;
    const uint64_t mask = __ballot(cand);
    const uint32_t nrec = (uint32_t)__builtin_popcountll(mask);
    if (nrec == 0) return;                                        // wave-uniform
    const uint4 rec = make_uint4(dm[0], dm[1], dm[2], dm[3]);
    if (nrec > kRawRecords) {
        a.spill[((size_t)tile * (2 * kEpoch) + eb * kEpoch + f) * 64 + lane] = rec;
        b.rawmask |= 1u << f;
        return;
    }
#if MI355_FUSED_ABLATE >= 4
    asm volatile("" ::"v"(rec.x), "v"(rec.y), "v"(rec.z), "v"(rec.w));
    return;
#endif
    if (cand) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        const uint32_t slot = (b.tail + rank) & (kRing - 1);
        w.rec[slot] = rec;
        w.rsrc[slot] = (uint16_t)(lane | (f << 6));
    }
    b.tail += nrec;
#if MI355_FUSED_ABLATE >= 3
    b.head = b.tail;
    return;
#endif
    if (b.tail - b.head >= 64) expand_round(a, w, b, eb, tile, 64, lane);
}

struct FrameGroup {
    uint4 c[kGroupFrames];
    // unconditional loads (frame index clamped): the number of vector-memory operations younger than any
    // load stays known at compile time (see diff_pack.hip, Group)
    __device__ __forceinline__ void load(const FusedArgs &a, uint32_t byte_off, int t0) {
        const int last = a.nframes - 1;
#pragma unroll
        for (int d = 0; d < kGroupFrames; d++) {
            const int t = min(t0 + d, last);
            const uint8_t *cb = uniform_ptr(a.cur + (size_t)t * a.stride);
            c[d] = load16<true, true>(cb + byte_off, 16);
        }
    }
};

// kGroupFrames frames of one tile, in pairs (the two byte counts share one DPP reduction).
__device__ __forceinline__ void fused_group(const FusedArgs &a, const FrameGroup &g, int t0, uint32_t f0,
                                            uint4 &st, bool keep, ThrConst tc, WaveLds &w, WaveBook &b,
                                            uint32_t eb, uint32_t tile, uint32_t lane) {
#pragma unroll
    for (int d = 0; d < kGroupFrames; d += 2) {
        const int t = t0 + d;
        uint32_t n0 = 0, n1 = 0;
        if (t < a.nframes) {   // wave-uniform
            uint32_t c0 = 24, c1 = 24;
            uint4 cur = g.c[d];
            if (!keep) cur = st;           // lanes beyond the end of the frame: no difference
            fused_step(a, cur, st, tc, w, b, eb, f0 + d, tile, lane, c0);
            if (t + 1 < a.nframes) {
                cur = g.c[d + 1];
                if (!keep) cur = st;
                fused_step(a, cur, st, tc, w, b, eb, f0 + d + 1, tile, lane, c1);
            }
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(
                wave_inclusive_scan((int)(c0 | (c1 << 16))), 63);
            n0 = ((tot & 0xffffu) - 64u * 24u) >> 2;
            n1 = ((tot >> 16) - 64u * 24u) >> 2;
        }
        w.n[eb][f0 + d] = n0;        // every lane stores the same value
        w.n[eb][f0 + d + 1] = n1;
    }
}

// The counts of epoch e leave for the other workgroups (see the head of the file).
// gsum word: bits 0..19 bytes (at most 64 workgroups x 4096), bits 20.. contributors.
// wgsum word: bits 0..15 bytes (at most 4096), bits 16..31 the launch tag (never 0; the buffer is cleared when
// the tag wraps), so a word left by an earlier launch is never taken for this one's.
// ready word (one per workgroup, own 128-byte line): tag << 16 | epochs every workgroup has published.
constexpr uint32_t kGsumOne = 1u << 20, kGsumMask = kGsumOne - 1u;
constexpr uint32_t kReadyStride = 32;

// Publication and arrival of one epoch by one wave of the workgroup.  Every step is a returning atomic at
// agent scope: its return means "performed", so the next step is ordered behind it without a release fence
// (which would write back this XCD's L2, full of everybody's fresh output lines).  gfx950 assumption, the same
// as k_scan_groups' ticket (tests/soak.py is its guard).
__device__ __forceinline__ void publish_and_arrive(const FusedArgs &a, const WaveLds *ws, uint32_t e, uint32_t eb,
                                                   uint32_t wg, uint32_t lane) {
    const uint32_t t = e * kEpoch + lane;
    const uint32_t group = wg / kWgPerGroup;
    uint32_t r0 = 0, r1 = 0;
    if (lane < (uint32_t)kEpoch && t < (uint32_t)a.nframes) {
        const uint32_t s = ws[0].n[eb][lane] + ws[1].n[eb][lane] + ws[2].n[eb][lane] + ws[3].n[eb][lane];
        r0 = __hip_atomic_exchange(&a.wgsum[(size_t)t * a.nwg + wg], s | (a.tag << 16), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        r1 = __hip_atomic_fetch_add(&a.gsum[(size_t)t * a.ngroups + group], s | kGsumOne, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::"v"(r0), "v"(r1) : "memory");     // the counts are performed
    uint32_t r = 0;
    if (lane == 0) r = __hip_atomic_fetch_add(&a.garrive[(size_t)e * a.ngroups + group], 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t members = min(kWgPerGroup, a.nwg - group * kWgPerGroup);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)r) != members - 1) return;   // wave-uniform
    // last of its group (24 workgroups per epoch at 1080p get here)
    r = 0;
    if (lane == 0) r = __hip_atomic_fetch_add(&a.arrive[e], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)r) != a.ngroups - 1) return;
    // every workgroup has published the epoch (max, not store: the next epoch's broadcast may overtake this one)
    const uint32_t v = (a.tag << 16) | (e + 1);
    for (uint32_t k = lane; k < a.nwg; k += 64)
        __hip_atomic_fetch_max(&a.ready[(size_t)k * kReadyStride], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Waits (bounded) for the workgroup's own flag: every workgroup has published epoch e.
__device__ __forceinline__ bool wait_epoch(const FusedArgs &a, uint32_t e, uint32_t wg) {
    const uint32_t *flag = &a.ready[(size_t)wg * kReadyStride];
    const uint32_t want = (a.tag << 16) | (e + 1);
    const uint64_t t0 = wall_clock64();
    for (uint32_t it = 1;; it++) {
        const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)agent_load(flag));
        if ((v >> 16) == a.tag && v >= want) return true;
        if (wall_clock64() - t0 > kSpinLimitTicks) return false;
        // the shared status word is looked at once in a while only (a word every waiting wave polls is a hot spot)
        if ((it & 1023u) == 0 && (uint32_t)__builtin_amdgcn_readfirstlane((int)agent_load(a.status)) != 0u) return false;
        __builtin_amdgcn_s_sleep(16);
    }
}

// What a wave asks for to place its workgroup in frames 2*wave and 2*wave+1 of an epoch: lane g < ngroups
// holds gsum[t][g], lane i holds wgsum[t][first workgroup of the group + i] for the workgroups before its own.
struct PrefixRows {
    uint32_t gs[2], ws[2];

    __device__ __forceinline__ void request(const FusedArgs &a, uint32_t epoch, uint32_t wave, uint32_t lane,
                                            uint32_t wg, uint32_t wg0) {
#pragma unroll
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t t = epoch * kEpoch + 2 * wave + k;
            gs[k] = ws[k] = 0;
            if (t < (uint32_t)a.nframes) {      // wave-uniform
                if (lane < a.ngroups) gs[k] = agent_load(&a.gsum[(size_t)t * a.ngroups + lane]);
                if (wg0 + lane < wg) ws[k] = agent_load(&a.wgsum[(size_t)t * a.nwg + wg0 + lane]);
            }
        }
    }

    // every word this wave needs carries its completeness: all contributors in, this launch's tag
    __device__ __forceinline__ bool complete(const FusedArgs &a, uint32_t epoch, uint32_t wave, uint32_t lane,
                                             uint32_t wg, uint32_t wg0) const {
        const uint32_t members = min(kWgPerGroup, a.nwg - min(a.nwg, lane * kWgPerGroup));
        bool ok = true;
#pragma unroll
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t t = epoch * kEpoch + 2 * wave + k;
            if (t < (uint32_t)a.nframes) {
                if (lane < a.ngroups && (gs[k] >> 20) != members) ok = false;
                if (wg0 + lane < wg && (ws[k] >> 16) != a.tag) ok = false;
            }
        }
        return __ballot(!ok) == 0;
    }
};

// A raw frame of the tile: its 64 records come back from the spill slot and are emitted at dst0.
__device__ __forceinline__ void emit_raw(const FusedArgs &a, uint32_t tile, uint32_t slot, uint32_t dst0,
                                         uint32_t lane) {
    const uint4 rec = a.spill[((size_t)tile * (2 * kEpoch) + slot) * 64 + lane];
    uint32_t m16 = record_map16(rec);
    const uint32_t cnt = (uint32_t)__builtin_popcount(m16);
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
    size_t dst = (size_t)dst0 + incl - cnt;
    const uint32_t xs0 = tile * kTileBytes + lane * 16u;
    if (__ballot(m16 != 0xffffu) == 0 && (size_t)dst0 + kTileBytes <= a.capacity) {
        // every byte of the tile changed: the indices are an arithmetic sequence, the differences the record
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t x = xs0 + 4 * k;
            u32x4 v = {x, x + 1, x + 2, x + 3};
            *reinterpret_cast<u32x4 *>(a.out_xs + dst + 4 * k) = v;
        }
        typedef uint32_t u32x4b __attribute__((ext_vector_type(4), aligned(1)));
        u32x4b dv = {rec.x, rec.y, rec.z, rec.w};
        *reinterpret_cast<u32x4b *>(a.out_diff + dst) = dv;
        return;
    }
    while (m16) {
        const int bit = __builtin_ctz(m16);
        m16 &= m16 - 1;
        const uint32_t dw = bit < 8 ? (bit < 4 ? rec.x : rec.y) : (bit < 12 ? rec.z : rec.w);
        if (dst < a.capacity) {
            a.out_xs[dst] = (int32_t)(xs0 + (uint32_t)bit);             // kernels.cu:315
            a.out_diff[dst] = (uint8_t)(dw >> (8 * (bit & 3)));        // kernels.cu:314
        }
        ++dst;
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_diff_fused(const FusedArgs a) {
    __shared__ WaveLds s_w[kWavesPerBlock];
    __shared__ uint32_t s_pref[kEpoch], s_tot[kEpoch];
    __shared__ uint32_t s_ok;        // 0: a bounded wait expired in this workgroup

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: keep it (and the tile) in SGPRs
    // Workgroups go to the 8 XCDs round-robin by linear id: XCD x takes the x-th eighth of the frame, so that
    // neighbouring tiles' output fragments meet in one L2.
    const uint32_t per_xcd = (a.nwg + 7u) / 8u;
    const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (wg >= a.nwg || (blockIdx.x >> 3) >= per_xcd) return;     // padding workgroups (whole workgroup)
    const uint32_t tile = wg * kWavesPerBlock + wave;
    const bool active = tile < a.ntiles;                          // wave-uniform
    const uint32_t tile_off = tile * kTileBytes;
    uint32_t byte_off = tile_off + lane * 16u;
    // a.n is a multiple of 16: a lane is inside the frame with all its bytes or with none
    const bool keep = active && byte_off < a.n;
    if (!keep) byte_off = 0;                                      // load something valid, ignore it
    const int T = a.nframes;
    const ThrConst tc{(127u - (uint32_t)a.thr) * 0x01010101u, (uint32_t)a.thr * 0x01010101u};
    WaveLds &w = s_w[wave];
    WaveBook b;

    uint4 st = load16<true>(a.state + byte_off, 16);
    const uint32_t group = wg / kWgPerGroup, wg0 = group * kWgPerGroup;
    uint32_t carry = 0;            // entries of all frames before the epoch being written out
    uint32_t prev_nfifo = 0, prev_raw = 0;
    const uint32_t nepochs = ((uint32_t)T + kEpoch - 1) / kEpoch;

    if (threadIdx.x == 0) s_ok = 1u;
    FrameGroup ga, gb;
    PrefixRows rows;
    rows.gs[0] = rows.gs[1] = rows.ws[0] = rows.ws[1] = 0;
    ga.load(a, byte_off, 0);
    for (uint32_t e = 0;; e++) {
        const uint32_t eb = e & 1u;
        const int t0 = (int)(e * kEpoch);
#pragma unroll 1
        for (int f0 = 0; f0 < kEpoch; f0 += 2 * kGroupFrames) {
            gb.load(a, byte_off, t0 + f0 + kGroupFrames);
            fused_group(a, ga, t0 + f0, (uint32_t)f0, st, keep, tc, w, b, eb, tile, lane);
            ga.load(a, byte_off, t0 + f0 + 2 * kGroupFrames);
            fused_group(a, gb, t0 + f0 + kGroupFrames, (uint32_t)(f0 + kGroupFrames), st, keep, tc, w, b, eb, tile, lane);
        }
        if (b.tail != b.head) expand_round(a, w, b, eb, tile, b.tail - b.head, lane);
        const uint32_t cur_nfifo = b.ftail, cur_raw = b.rawmask;
        b.ftail = 0;
        b.rawmask = 0;
        lds_barrier();                                     // the four waves' n[eb][*] are in LDS
#if MI355_FUSED_ABLATE == 0
        if (wave == (e & 3u)) publish_and_arrive(a, s_w, e, eb, wg, lane);
#endif

        // ---- write-out of the previous epoch (of this one, after the last) ----
        for (uint32_t pass = 0; pass < 2; pass++) {
            const bool last = e + 1 == nepochs;
            if (pass == 0 && e == 0) continue;
            if (pass == 1 && !last) break;
            const uint32_t we = pass == 0 ? e - 1 : e, wb = we & 1u;
            const uint32_t nfifo = pass == 0 ? prev_nfifo : cur_nfifo;
            const uint32_t raw = pass == 0 ? prev_raw : cur_raw;
#if MI355_FUSED_ABLATE == 0
            // prefix of this workgroup in frames 2*wave, 2*wave+1 of the epoch: wait for our own flag (never
            // poll the shared rows), then ask for them; the words still say themselves whether they are complete
            for (;;) {
                if (!wait_epoch(a, we, wg)) {
                    __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_ok = 0u;
                    break;
                }
                rows.request(a, we, wave, lane, wg, wg0);
                if (rows.complete(a, we, wave, lane, wg, wg0)) break;
                __builtin_amdgcn_s_sleep(64);     // flag up but a word not there: cannot happen; do not spin hot
            }
#pragma unroll
            for (uint32_t k = 0; k < 2; k++) {
                const uint32_t g = rows.gs[k] & kGsumMask;
                s_pref[2 * wave + k] = wave_sum((rows.ws[k] & 0xffffu) + (lane < group ? g : 0u));
                s_tot[2 * wave + k] = wave_sum(g);
            }
            lds_barrier();
            if (!s_ok) return;                              // every wave of the workgroup reads the same word
#else
            if (lane < (uint32_t)kEpoch) { s_pref[lane] = 0; s_tot[lane] = 0; }
            lds_barrier();
#endif
            // lanes 0..kEpoch-1: where this tile's entries of frame `lane` go
            uint32_t tot_f = 0, dst_f = 0, n_f = 0, mine = 0;
            if (lane < (uint32_t)kEpoch) {
                tot_f = s_tot[lane];
                dst_f = s_pref[lane];
                for (uint32_t k = 0; k < kWavesPerBlock; k++) {
                    const uint32_t v = s_w[k].n[wb][lane];
                    if (k < wave) dst_f += v;
                    if (k == wave) mine = v;
                }
                n_f = ((raw >> lane) & 1u) ? 0u : mine;
            }
            const uint32_t tot_incl = (uint32_t)wave_inclusive_scan((int)tot_f);
            const uint32_t n_incl = (uint32_t)wave_inclusive_scan((int)n_f);
            dst_f += carry + tot_incl - tot_f;                           // global index of the frame's first entry of this tile
            const uint32_t dm_f = dst_f - (n_incl - n_f);                // ... minus the entry's FIFO position
            if (wg == 0 && wave == 0 && lane < (uint32_t)kEpoch && we * kEpoch + lane < (uint32_t)T)
                a.offsets[we * kEpoch + lane] = carry + tot_incl - tot_f;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)tot_incl, 63);
            if (last && pass == 1 && wg == 0 && wave == 0 && lane == 0) a.offsets[T] = carry;
#if MI355_FUSED_ABLATE < 2
            if (active) {
                for (uint32_t j0 = 0; j0 < nfifo; j0 += 64) {
                    const uint32_t j = j0 + lane;
                    uint32_t v = 0;
                    if (j < nfifo) v = j < kFifo ? w.fifo[wb][j] : a.ovf[((size_t)tile * 2 + wb) * kOvf + (j - kFifo)];
                    const uint32_t f = (v >> 10) & 7u;
                    // 32-bit sum: dm_f is "frame's first entry minus its FIFO position" modulo 2^32
                    const size_t dst = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(f * 4u), (int)dm_f) + j;
                    if (j < nfifo && dst < a.capacity) {
                        a.out_xs[dst] = (int32_t)(tile_off + (v & 1023u));      // kernels.cu:315
                        a.out_diff[dst] = (uint8_t)(v >> 16);                   // kernels.cu:314
                    }
                }
                for (uint32_t m = raw; m; m &= m - 1) {
                    const uint32_t f = (uint32_t)__builtin_ctz(m);
                    emit_raw(a, tile, wb * kEpoch + f, (uint32_t)__builtin_amdgcn_readlane((int)dst_f, (int)f), lane);
                }
            }
#endif
            lds_barrier();   // s_pref / s_tot / n[wb] are free again
        }
        prev_nfifo = cur_nfifo;
        prev_raw = cur_raw;
        if (e + 1 == nepochs) break;
    }
    if (keep) *reinterpret_cast<uint4 *>(a.state + byte_off) = st;
}

uint32_t fused_groups(uint32_t nwg) { return (nwg + kWgPerGroup - 1) / kWgPerGroup; }
size_t fused_spill_records(uint32_t ntiles) { return (size_t)ntiles * 2 * kEpoch * 64; }
size_t fused_ovf_entries(uint32_t ntiles) { return (size_t)ntiles * 2 * kOvf; }
uint32_t fused_epochs(int nframes) { return ((uint32_t)nframes + kEpoch - 1) / kEpoch; }
size_t fused_ready_words(uint32_t nwg) { return (size_t)nwg * kReadyStride; }

// Workgroups of k_diff_fused the device keeps resident at once (0 on error).
uint32_t fused_capacity(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_diff_fused, 64 * kWavesPerBlock, 0) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu > 0 && cus > 0 ? (uint32_t)per_cu * (uint32_t)cus : 0;
}

hipError_t launch_diff_fused(const FusedArgs &a, hipStream_t s) {
    const uint32_t per_xcd = (a.nwg + 7u) / 8u;
    hipLaunchKernelGGL(k_diff_fused, dim3(per_xcd * 8u), dim3(64 * kWavesPerBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace mi355
