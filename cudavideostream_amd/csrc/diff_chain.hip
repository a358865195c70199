// diff_chain.hip -- EXPERIMENT (opt-in, MI355_FLAG_CHAIN): the stateless (pair) form of diff + threshold +
// ordered pack in ONE pass over the frames.  Bit-exact with the log path (same tests).  On the MI355X
// (profiles/README.md "r02 chained pair kernel"): dense pairs as fast as the log path (S0 0.336 vs 0.338 ms per
// 32 frames), sparse pairs slower (1.13 vs 0.61 ms per 256 frames): of its 1.13 ms, 0.47 is the pass itself,
// 0.29 the look-back (the persistent workgroups reach it in lockstep, the chip idles meanwhile) and 0.38 the
// emission (tile-parallel: a step with 23 entries costs as much as one with 1000).  Kept as the measured
// answer to "single-pass compaction for the pair form", not as the product path.
//
// Frame pairs carry no state from frame to frame (tests/algorithms_benchmarks.cu style independent pairs,
// BASELINE config 5's round-robin frames), so the batch is one long ordered compaction: it is done the way
// a single-pass stream compaction is done, with a chained scan (decoupled look-back), and moves exactly the
// algorithmic bytes: 2N read + 5P written per frame -- no record log, no second kernel.  The three-kernel log
// path (diff_pack.hip) reads 2N, writes and re-reads a 16-byte record per candidate lane and then writes 5P.
//
//   * a workgroup (4 waves) owns one *block* = 32 consecutive 1 KiB tiles of one frame; a wave owns 8 of them
//     and parks their masked differences in LDS (8 steps x 1 KiB) while the block's place in the output is
//     found (in registers the 16 statically indexed copies of the emission code cost more than they save);
//   * chain inside a frame: a block publishes its flagged-byte count (aggregate), looks back over the
//     descriptors of the blocks before it in the same frame (64 per load) until it meets one that already
//     knows its inclusive prefix, and publishes its own inclusive prefix;
//   * chain over frames, the same way: the last block of frame t publishes the frame's total, looks back over
//     the frames before it and publishes the running total (this is also offsets[t+1]); every block looks back
//     over the frame descriptors for the entries of the frames before its own;
//   * emission: the wave walks its 8 steps again, from LDS: 16-bit map of flagged bytes per lane, DPP
//     scan, entries staged in LDS in output order, coalesced stores (the expander's staging, fed from
//     registers); a step in which every byte changed skips the stage: indices are an arithmetic sequence.
//
// Descriptors are 64-bit words {value:32 | launch tag:16 | state:2}; a word left by an earlier launch has another
// tag and reads as "not there yet" (the buffer is cleared when the tag wraps).  The grid is persistent: as many
// workgroups as the device keeps resident (chain_capacity), workgroup w takes blocks w, w + grid, ... with the
// loads of its next block already in flight while it places and writes out the current one.  Blocks depend on
// blocks with a lower number only and every workgroup is resident, so whoever is waited for is running.  Every
// wait is bounded (kChainSpinLimit): on expiry the kernel raises status[0] and the workgroup leaves, so the grid
// always drains.
#include "pack_common.h"

namespace mi355 {

#ifndef MI355_CHAIN_ABLATE
#define MI355_CHAIN_ABLATE 0   // timing builds only: 1 = no emission, 2 = also no look-back (every block at 0)
#endif
constexpr int kSteps = 8;                          // tiles per wave
constexpr uint32_t kBlockTiles = kSteps * kWavesPerBlock;   // 32 tiles = 32 KiB of a frame per workgroup
constexpr uint32_t kStage = 1024;                  // entries staged per wave = the most one step can hold
constexpr uint64_t kChainSpinLimit = 200000000ull; // 2 s of the 100 MHz wall clock
constexpr uint64_t kAgg = 1, kIncl = 2;

__device__ __forceinline__ uint64_t desc_word(uint32_t value, uint32_t tag, uint64_t state) {
    return (uint64_t)value | ((uint64_t)tag << 32) | (state << 48);
}
__device__ __forceinline__ uint32_t desc_state(uint64_t d, uint32_t tag) {   // 0 = not there (yet)
    return (uint32_t)((d >> 32) & 0xffffu) == tag ? (uint32_t)(d >> 48) & 3u : 0u;
}
__device__ __forceinline__ uint64_t agent_load64(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_store64(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane(wave_inclusive_scan((int)v), 63);
}
__device__ __forceinline__ void lds_barrier_c() {   // LDS-only workgroup barrier (no wait for loads in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void wave_lds_handoff_c() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Decoupled look-back over row[0..hi): the sum of the values of all entries, taken from the nearest entry that
// already knows its inclusive prefix and the aggregates after it.  kWin windows of 64 descriptors are requested
// at once (lane 0 of window 0 = nearest): in a persistent grid the blocks of a round publish their aggregates at
// about the same time, so a block usually has to go all the way back to the first block of its frame.
// Waits only for entries that have not published anything yet.  One wave; false on timeout.
constexpr int kWin = 3;
__device__ __forceinline__ bool look_back(const uint64_t *row, uint32_t hi, uint32_t tag, uint32_t lane, uint32_t &excl) {
    uint32_t sum = 0;
    const uint64_t t0 = wall_clock64();
    while (hi > 0) {
        uint64_t d[kWin];
#pragma unroll
        for (int w = 0; w < kWin; w++) {
            const uint32_t back = 64u * w + lane + 1u;               // distance from hi
            d[w] = back <= hi ? agent_load64(&row[hi - back]) : 0ull;
        }
        bool again = false, done = false;
#pragma unroll
        for (int w = 0; w < kWin; w++) {
            if (again || done || hi == 0) continue;                  // wave-uniform
            const uint32_t nv = min(64u, hi);
            const uint32_t st = desc_state(d[w], tag);
            const uint64_t missing = __ballot(lane < nv && st == 0u);
            const uint64_t incl = __ballot(lane < nv && st == (uint32_t)kIncl);
            const uint32_t first_incl = incl ? (uint32_t)__builtin_ctzll(incl) : 64u;
            const uint32_t upto = min(first_incl + 1u, nv);          // lanes [0, upto) are summed
            const uint64_t need = upto >= 64u ? ~0ull : ((1ull << upto) - 1ull);
            if (missing & need) { again = true; continue; }          // a predecessor has not published yet
            sum += wave_sum32(lane < upto ? (uint32_t)d[w] : 0u);
            if (first_incl < nv) done = true;                        // met an inclusive prefix
            else hi -= nv;
        }
        if (done) break;
        if (again) {
            if (wall_clock64() - t0 > kChainSpinLimit) return false;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    excl = sum;
    return true;
}

// Entries staged at s_xs/s_df[0..count) leave for out[first ...] with coalesced stores (see flush_entries).
__device__ __forceinline__ void flush_stage(const ChainArgs &a, const uint16_t *s_xs, const uint8_t *s_df,
                                            size_t first, uint32_t count, uint32_t xs0, uint32_t lane) {
    wave_lds_handoff_c();
    const uint32_t n = first >= a.capacity ? 0u : (uint32_t)(a.capacity - first < count ? a.capacity - first : count);
    int32_t *xsp = a.out_xs + first;
    uint8_t *dfp = a.out_diff + first;
    for (uint32_t i = lane; i < n; i += 64) xsp[i] = (int32_t)(xs0 + s_xs[i]);        // kernels.cu:315
    const uint32_t lead = (4u - (uint32_t)((uintptr_t)dfp & 3u)) & 3u;
    const uint32_t head = lead < n ? lead : n;
    const uint32_t body = (n - head) >> 2;
    if (lane < head) dfp[lane] = s_df[lane];                                            // kernels.cu:314
    for (uint32_t k = lane; k < body; k += 64) {
        const uint8_t *q = s_df + head + 4 * k;
        const uint32_t v = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        *reinterpret_cast<uint32_t *>(dfp + head + 4 * (size_t)k) = v;
    }
    const uint32_t tail = head + 4 * body + lane;
    if (tail < n) dfp[tail] = s_df[tail];
    wave_lds_handoff_c();
}

// Where a wave finds one block: frame bases and its first tile.
struct BlockAt {
    const uint8_t *cur, *prv;
    uint32_t t, g, tile0;
};

__global__ __launch_bounds__(256) void k_diff_pairs_chained(const ChainArgs a) {
    __shared__ uint4 s_dm[kWavesPerBlock][kSteps][64];
    __shared__ uint16_t s_xs[kWavesPerBlock][kStage];
    __shared__ uint8_t s_df[kWavesPerBlock][kStage];
    __shared__ uint32_t s_wtot[kWavesPerBlock];
    __shared__ uint32_t s_base, s_ok;

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const ThrConst tc{(127u - (uint32_t)a.thr) * 0x01010101u, (uint32_t)a.thr * 0x01010101u};
    const uint32_t nblocks = (uint32_t)a.nframes * a.ngroups;
    const uint32_t last_tile = a.ntiles - 1u;
    uint4 (*dm)[64] = s_dm[wave];
    uint16_t *sx = s_xs[wave];
    uint8_t *sd = s_df[wave];
    if (threadIdx.x == 0) s_ok = 1u;

    auto locate = [&](uint32_t b) {
        BlockAt at;
        b = min(b, nblocks - 1u);     // past the end: the last block again (loads that are never looked at)
        at.t = b / a.ngroups;
        at.g = b - at.t * a.ngroups;
        at.tile0 = at.g * kBlockTiles + wave * kSteps;
        at.cur = uniform_ptr(a.cur + (size_t)at.t * a.stride);
        at.prv = uniform_ptr(a.prev + (size_t)at.t * a.stride);
        return at;
    };
    // Loads are unconditional (tile index clamped, lanes beyond the frame read its first bytes and are ignored)
    // and run kAhead steps ahead of the arithmetic -- across block boundaries, so that a workgroup's loads of its
    // NEXT block are in flight while it finds the place of this one and writes it out.  Slot = step % kSlots,
    // static because kSteps is a multiple of kSlots.
    constexpr int kSlots = 8, kAhead = 7;
    static_assert(kSteps % kSlots == 0 && kAhead < kSlots, "register slots are indexed statically");
    uint4 c[kSlots], p[kSlots];
    auto issue = [&](const BlockAt &at, int s) {
        const uint32_t tile = min(at.tile0 + (uint32_t)s, last_tile);
        uint32_t off = tile * kTileBytes + lane * 16u;
        if (off >= a.n) off = 0;                 // a.n is a multiple of 16
        c[s % kSlots] = load16<true, true>(at.cur + off, 16);
        p[s % kSlots] = load16<true, true>(at.prv + off, 16);
    };

    // Workgroup w takes blocks w, w + gridDim.x, ...: every block depends on lower-numbered blocks only, and all
    // workgroups are resident (the grid is sized to the device), so whoever is waited for is running.
    uint32_t b = blockIdx.x;
    BlockAt at = locate(b);
#pragma unroll
    for (int s = 0; s < kAhead; s++) issue(at, s);
    while (b < nblocks) {
        const BlockAt nx = locate(b + gridDim.x);
        const uint32_t t = at.t, g = at.g, tile0 = at.tile0;
        const uint32_t nst = tile0 < a.ntiles ? min((uint32_t)kSteps, a.ntiles - tile0) : 0u;   // wave-uniform

        // ---- pass over the block: masked differences into LDS, flagged bytes per lane ----
        uint32_t acc = 0;    // sum over the steps of 4 * flags + 24
#pragma unroll
        for (int s = 0; s < kSteps; s++) {
            if (s + kAhead < kSteps) issue(at, s + kAhead);
            else issue(nx, s + kAhead - kSteps);
            const bool inside = (uint32_t)s < nst && (tile0 + (uint32_t)s) * kTileBytes + lane * 16u < a.n;
            const uint4 cv = c[s % kSlots], pv = p[s % kSlots];
            const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w}, sw[4] = {pv.x, pv.y, pv.z, pv.w};
            uint32_t d4[4], sel[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t x;
                uint32_t fh = dword_flags(cw[k], sw[k], tc, x);
                if (!inside) fh = 0;                                   // beyond the frame / the block: nothing flagged
                sel[k] = perm_select(fh);
                d4[k] = __builtin_amdgcn_perm(bytes_sub_from_x(cw[k], sw[k], x), 0u, sel[k]);   // 0 where un-flagged
            }
            dm[s][lane] = make_uint4(d4[0], d4[1], d4[2], d4[3]);
            acc += __builtin_amdgcn_sad_u8(sel[0] + sel[1] + sel[2] + sel[3], 0u, 0u);
            __builtin_amdgcn_sched_barrier(0);    // keep the load order (the scheduler would hoist every load)
        }
        const uint32_t lane_total = (acc - (uint32_t)kSteps * 24u) >> 2;
        const uint32_t wtot = wave_sum32(lane_total);

        // ---- the block's place in the output ----
        if (lane == 0) s_wtot[wave] = wtot;
        lds_barrier_c();
        uint32_t wexcl = 0, agg = 0;
#pragma unroll
        for (uint32_t k = 0; k < kWavesPerBlock; k++) {
            const uint32_t v = s_wtot[k];
            if (k < wave) wexcl += v;
            agg += v;
        }
        if (wave == 0 && MI355_CHAIN_ABLATE >= 2) {
            if (lane == 0) { s_base = agg; a.offsets[t] = agg; a.offsets[t + 1] = agg; agent_store64(a.desc + (size_t)t * a.ngroups + g, desc_word(agg, a.tag, kIncl)); }
        } else if (wave == 0) {
            // two chains of the same kind: blocks inside the frame, frames inside the batch.  Nobody waits for an
            // inclusive prefix to be handed down (a chain of hand-overs costs two memory latencies per link):
            // aggregates are enough.
            const uint64_t *row = a.desc + (size_t)t * a.ngroups;
            uint64_t *mine = a.desc + (size_t)t * a.ngroups + g;
            const bool last = g == a.ngroups - 1;
            if (lane == 0) agent_store64(mine, desc_word(agg, a.tag, g == 0 ? kIncl : kAgg));
            uint32_t excl = 0, fbase = 0;
            bool ok = look_back(row, g, a.tag, lane, excl);
            if (ok && g != 0 && lane == 0) agent_store64(mine, desc_word(excl + agg, a.tag, kIncl));
            const uint32_t ftotal = excl + agg;      // of the whole frame, if this is its last block
            if (ok && last && t != 0 && lane == 0) agent_store64(&a.fdesc[t], desc_word(ftotal, a.tag, kAgg));
            if (ok) ok = look_back(a.fdesc, t, a.tag, lane, fbase);     // entries of the frames before this one
            if (ok) {
                if (g == 0 && lane == 0) a.offsets[t] = fbase;
                if (last && lane == 0) {
                    agent_store64(&a.fdesc[t], desc_word(fbase + ftotal, a.tag, kIncl));
                    a.offsets[t + 1] = fbase + ftotal;
                }
                if (lane == 0) s_base = fbase + excl;
            } else if (lane == 0) {
                __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ok = 0u;
            }
        }
        lds_barrier_c();
        if (!s_ok) return;

        // ---- emission from LDS ----
        const size_t dst0 = (size_t)s_base + wexcl;
        const uint32_t xs0 = tile0 * kTileBytes;
        uint32_t carry = 0, flushed = 0;     // entries emitted / already stored
        auto emit = [&](const uint4 rec, const uint32_t s) {
            uint32_t m16 = record_map16(rec);
            const uint32_t cnt = (uint32_t)__builtin_popcount(m16);
            if (__ballot(m16 != 0xffffu) == 0) {
                // every byte of the tile changed: indices are an arithmetic sequence, the differences the record
                if (carry != flushed) { flush_stage(a, sx, sd, dst0 + flushed, carry - flushed, xs0, lane); flushed = carry; }
                const size_t d = dst0 + carry + lane * 16u;
                const uint32_t x = xs0 + s * kTileBytes + lane * 16u;
                if (dst0 + carry + kTileBytes <= a.capacity) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
                    typedef uint32_t u32x4b __attribute__((ext_vector_type(4), aligned(1)));
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const u32x4 v = {x + 4 * k, x + 4 * k + 1, x + 4 * k + 2, x + 4 * k + 3};
                        *reinterpret_cast<u32x4 *>(a.out_xs + d + 4 * k) = v;
                    }
                    const u32x4b dv = {rec.x, rec.y, rec.z, rec.w};
                    *reinterpret_cast<u32x4b *>(a.out_diff + d) = dv;
                } else {
                    const uint32_t w4[4] = {rec.x, rec.y, rec.z, rec.w};
#pragma unroll
                    for (uint32_t k = 0; k < 16; k++)
                        if (d + k < a.capacity) {
                            a.out_xs[d + k] = (int32_t)(x + k);
                            a.out_diff[d + k] = (uint8_t)(w4[k >> 2] >> (8 * (k & 3)));
                        }
                }
                carry += kTileBytes;
                flushed = carry;
                return;
            }
            const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0) return;
            if (carry - flushed + total > kStage) { flush_stage(a, sx, sd, dst0 + flushed, carry - flushed, xs0, lane); flushed = carry; }
            uint32_t e = carry - flushed + incl - cnt;
            carry += total;
            const uint32_t src16 = s * kTileBytes + lane * 16u;
            while (m16) {
                const int bit = __builtin_ctz(m16);
                m16 &= m16 - 1;
                const uint32_t dw = bit < 8 ? (bit < 4 ? rec.x : rec.y) : (bit < 12 ? rec.z : rec.w);
                sx[e] = (uint16_t)(src16 + (uint32_t)bit);
                sd[e] = (uint8_t)(dw >> (8 * (bit & 3)));
                ++e;
            }
        };
        if (agg && MI355_CHAIN_ABLATE == 0) {     // workgroup-uniform: nothing flagged in the whole block is the common case of a still scene
#pragma unroll 1
            for (uint32_t s = 0; s < nst; s++) emit(dm[s][lane], s);
            if (carry != flushed) flush_stage(a, sx, sd, dst0 + flushed, carry - flushed, xs0, lane);
        }
        b += gridDim.x;
        at = nx;
        // s_wtot / s_base are rewritten only after the next block's first barrier, which every wave reaches
        // after it has read them here
    }
}

uint32_t chain_groups(uint32_t ntiles) { return (ntiles + kBlockTiles - 1) / kBlockTiles; }

// Workgroups of k_diff_pairs_chained the device keeps resident at once (0 on error).
uint32_t chain_capacity(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_diff_pairs_chained, 64 * kWavesPerBlock, 0) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu > 0 && cus > 0 ? (uint32_t)per_cu * (uint32_t)cus : 0;
}

hipError_t launch_diff_chain(const ChainArgs &a, uint32_t resident, hipStream_t s) {
    const uint32_t nblocks = (uint32_t)a.nframes * a.ngroups;
    hipLaunchKernelGGL(k_diff_pairs_chained, dim3(nblocks < resident ? nblocks : resident), dim3(64 * kWavesPerBlock), 0, s, a);
    return hipGetLastError();
}

}  // namespace mi355
