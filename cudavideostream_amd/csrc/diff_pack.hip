// diff_pack.hip -- K1: per-byte |cur - state| > T, negative feedback, ordered sparse pack (gfx950).
//
// Replaces kernel2 (reference server/src/kernels.cu:289-334, launched <<<1,1024>>> at :505) and
// reproduces the CPU statement of the same step (tests/cuda_streaming/test.cu:560-576): ascending
// byte order, bounded, no atomics.
//
// The path is byte streaming, but on gfx950 an integer VALU instruction occupies a SIMD for 4 cycles
// per wave64, so at the HBM rate a wave has only ~100 VALU instructions per KiB: the kernel is built
// around *bytes per instruction*, not around the memory system alone.
//
//   k_diff_pack : one wave64 owns one 1 KiB tile of the frame (lane l = bytes 16l..16l+15, one
//                 global_load_dwordx4 per lane per frame) for ALL frames of the batch.  In stream mode
//                 the tile's state lives in 4 VGPRs per lane across the batch, so a frame costs N
//                 bytes of HBM reads instead of 2N; frames are double buffered in two register groups.
//                 Per frame and dword, 4 bytes at a time (SWAR + v_bitop3 + v_perm):
//                   flags   9 ops  exact 9-bit compare of 4 bytes (dword_flags)
//                   select  2 ops  v_perm selector from the flags
//                   state   1 op   v_perm(cur, state)          negative feedback
//                   diff    3 ops  per-byte (cur - state), zeroed where un-flagged
//                   count   3 ops per 16 bytes (sum of the four selectors, v_sad_u8)
//                 A lane with at least one flagged byte is a *candidate*; candidates are ranked with
//                 one ballot + mbcnt and each stores its 16 masked diff bytes as ONE 16-byte record
//                 (un-flagged bytes are 0; a flagged byte is never 0 because |df| > T >= 0).  Records
//                 of a tile are appended to a chunk-interleaved log (consecutive frames fill whole
//                 cache lines; all tiles' current chunks are neighbours in memory).  Per (frame, tile)
//                 one 16-byte meta word keeps {candidate ballot, flagged-byte count, log position}.
//   k_scan_groups: per frame, flagged bytes before every group of 64 tiles; its last workgroup scans the
//               frame totals into offsets[0..T].
//   k_expand    : one wave per (frame, 16 tiles): turns records into the caller's packed, frame-major,
//                 ascending (xs, diff) arrays -- or the socket's byte stream -- through an LDS stage
//                 and coalesced stores.
// No spin waits; the only inter-workgroup communication is the completion ticket of k_scan_groups;
// results are independent of dispatch order.
#include "pack_common.h"

namespace mi355 {

// Ablation builds (tools/ablate.sh, never shipped): 1 = no record stores, 2 = also no meta stores,
// 3 = loads + state fold only (memory floor of the stream loop), 4 / 5 = record store of one lane /
// of all 64 lanes (bytes vs instruction cost).  0 = the product.
#ifndef MI355_ABLATE
#define MI355_ABLATE 0
#endif

#ifndef MI355_K1_PREFETCH
#define MI355_K1_PREFETCH 4
#endif
constexpr int kPrefetch = MI355_K1_PREFETCH;  // frames per register group (two groups: 8 x 1 KiB in flight per wave)
static_assert(kPrefetch % 2 == 0, "frames are processed in pairs");

// Record log: position `pos` of tile `tile` lives at record index
//   ((pos / 64) * ntiles + tile) * 64 + pos % 64
// (fits 32 bits: T * W * 64 = max_batch * N / 16 < 2^28; chunk and W are below 2^24)
__device__ __forceinline__ uint32_t rec_index(uint32_t pos, uint32_t tile, uint32_t ntiles) {
    return (__umul24(pos >> 6, ntiles) + tile) * 64u + (pos & 63u);
}

// One frame of one tile: compare, update the state, append the candidate records at log position
// `run`.  Returns the candidate ballot; cnt4 receives 4 * (this lane's flagged bytes) + 24.
__device__ __forceinline__ uint64_t pack_step(const uint4 c, uint4 &s, ThrConst tc, uint4 *rec_log,
                                              uint32_t run, uint32_t tile, uint32_t ntiles,
                                              uint32_t &cnt4) {
#if MI355_ABLATE == 3
    s.x ^= c.x; s.y ^= c.y; s.z ^= c.z; s.w ^= c.w;
    cnt4 = 24;
    return 0;
#endif
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
    uint32_t sw[4] = {s.x, s.y, s.z, s.w};
    uint32_t dm[4], sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t x;
        const uint32_t fh = dword_flags(cw[k], sw[k], tc, x);
        sel[k] = perm_select(fh);
        const uint32_t d = bytes_sub_from_x(cw[k], sw[k], x);
        dm[k] = __builtin_amdgcn_perm(d, 0u, sel[k]);            // diff where flagged, 0 elsewhere
        // negative feedback (kernels.cu:316-331): flagged bytes take the current value, the others
        // keep the previous one -> the state is the frame the client reconstructs
        sw[k] = __builtin_amdgcn_perm(cw[k], sw[k], sel[k]);
    }
    s = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    // every selector byte is j + 4*flag: the byte sum of the four selectors is 4*flags + 4*(0+1+2+3)
    cnt4 = __builtin_amdgcn_sad_u8(sel[0] + sel[1] + sel[2] + sel[3], 0u, 0u);

    const bool cand = ((dm[0] | dm[1]) | (dm[2] | dm[3])) != 0u;
    const uint64_t mask = __ballot(cand);
#if MI355_ABLATE >= 1 && MI355_ABLATE < 4
    asm volatile("" ::"v"(dm[0]), "v"(dm[1]), "v"(dm[2]), "v"(dm[3]));
#else
#if MI355_ABLATE == 4   // same instruction stream, 1/19 of the bytes: only the first candidate stores
    if (cand && __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)) == 0) {
#elif MI355_ABLATE == 5 // every lane stores (1 KiB per step)
    if (true) {
#else
    if (cand) {
#endif
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        rec_log[rec_index(run + rank, tile, ntiles)] = make_uint4(dm[0], dm[1], dm[2], dm[3]);
    }
#endif
    return mask;
}

// A group = kPrefetch consecutive frames of one tile held in registers.  Loads are always issued
// (frame index clamped to T-1) so that the number of vector-memory operations younger than any
// load is known at compile time: on gfx950 loads and stores share one in-order vmcnt, and a
// conditional load would make the compiler fall back to s_waitcnt vmcnt(0) -- i.e. no prefetch.
template <bool PAIR, bool FAST>
struct Group {
    uint4 c[kPrefetch];
    uint4 p[kPrefetch];

    // The frame base (a.cur + t*stride) is wave-uniform and the lane's byte offset is a 32-bit VGPR:
    // the loads use the SGPR-base + VGPR-offset form and need no per-lane 64-bit address arithmetic.
    __device__ __forceinline__ void load(const PackArgs &a, uint32_t byte_off, int t0, int valid) {
        const int last = a.nframes - 1;
#pragma unroll
        for (int d = 0; d < kPrefetch; d++) {
            const int t = min(t0 + d, last);
            const uint8_t *cb = uniform_ptr(a.cur + (size_t)t * a.stride);
            c[d] = load16<FAST, !PAIR>(cb + byte_off, valid);   // stream frames: read once
            if (PAIR) {
                const uint8_t *pb = uniform_ptr(a.prev + (size_t)t * a.stride);
                p[d] = load16<FAST>(pb + byte_off, valid);
            }
        }
    }
};

template <bool PAIR, bool FAST>
__device__ __forceinline__ void pack_group(const PackArgs &a, const Group<PAIR, FAST> &g, int t0,
                                           uint4 &st, uint32_t &run, uint32_t tile, ThrConst tc,
                                           int lane) {
    // The group's kPrefetch meta words are assembled in lanes 0..kPrefetch-1 and leave with ONE store.
    uint4 meta = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < kPrefetch; d += 2) {
        const int t = t0 + d;
        if (t >= a.nframes) break;  // wave-uniform
        const bool two = t + 1 < a.nframes;
        // byte counts of the two frames share one register (16-bit fields: a wave total of
        // 4*flags + 24 per lane is at most 64 * 88 = 5632) and one DPP reduction
        uint32_t c0 = 24, c1 = 24;
        if (PAIR) st = g.p[d];
        const uint32_t run0 = run;
        const uint64_t m0 = pack_step(g.c[d], st, tc, a.rec, run, tile, a.ntiles, c0);
        run += (uint32_t)__builtin_popcountll(m0);
        const uint32_t run1 = run;
        uint64_t m1 = 0;
        if (two) {
            if (PAIR) st = g.p[d + 1];
            m1 = pack_step(g.c[d + 1], st, tc, a.rec, run, tile, a.ntiles, c1);
            run += (uint32_t)__builtin_popcountll(m1);
        }
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(
            wave_inclusive_scan((int)(c0 | (c1 << 16))), 63);
        const uint32_t n0 = ((tot & 0xffffu) - 64u * 24u) >> 2, n1 = ((tot >> 16) - 64u * 24u) >> 2;
        // all four values are wave-uniform: v_writelane drops them into lanes d and d+1
        write_lane(meta.x, (uint32_t)m0, d);
        write_lane(meta.y, (uint32_t)(m0 >> 32), d);
        write_lane(meta.z, n0, d);
        write_lane(meta.w, run0, d);
        write_lane(meta.x, (uint32_t)m1, d + 1);
        write_lane(meta.y, (uint32_t)(m1 >> 32), d + 1);
        write_lane(meta.z, n1, d + 1);
        write_lane(meta.w, run1, d + 1);
    }
    if (lane < kPrefetch && t0 + lane < a.nframes && MI355_ABLATE != 2 && MI355_ABLATE != 3)
        a.meta[(size_t)(t0 + lane) * a.ntiles + tile] = meta;
}

// ---- the steady state of the stream loop, written for exact s_waitcnt counts ---------------------------
// gfx950 has ONE in-order vmcnt for loads and stores.  The compiler can only wait for "all but the N youngest"
// operations, and N must hold on every path that reaches the wait: a store under a divergent `if` (skipped with
// s_cbranch_execz when no lane takes it) or a frame loop with `break`s makes N collapse -- the first form of this
// kernel ended up with `s_waitcnt vmcnt(0)` at the head of its loop (every 8 frames each wave drained its prefetch
// AND waited for the acknowledgement of the store it had just issued) and with waits that covered the previous
// group's record stores in front of every frame (ISA in profiles/r03_k1_waitcnt.md).  Here
//   * the main loop handles only complete groups: no frame-count conditionals inside,
//   * every record / meta store is a raw buffer store that is ALWAYS issued; lanes with nothing to store carry an
//     offset beyond the descriptor's range and the hardware drops them (no branch, no traffic),
//   * frame loads are raw buffer loads through a descriptor rebased per frame (SGPR arithmetic only),
// so the number of vector-memory operations between a load and its first use is a compile-time constant and the
// waits are exact: a frame's loads are only ever waited for with the stores of the group before still in flight.
#ifndef MI355_K1V
#define MI355_K1V 2
#endif
#ifndef MI355_K1_PIN_LOADS
#define MI355_K1_PIN_LOADS 1
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kRsrcWord3 = 0x00020000u;   // raw buffer, 32-bit data format (gfx9 family)
constexpr uint32_t kOOB = 0xFFFFFFFFu;         // beyond every descriptor's num_records: the lane's store is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, (int)kRsrcWord3);
}

template <bool PAIR>
struct Group2 {
    uint4 c[kPrefetch];
    uint4 p[kPrefetch];
    __device__ __forceinline__ void load(const PackArgs &a, uint32_t voff, int t0) {
        const int last = a.nframes - 1;
#pragma unroll
        for (int d = 0; d < kPrefetch; d++) {
            const int t = min(t0 + d, last);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.cur + (size_t)t * a.stride, a.n), voff, 0,
                                                                  PAIR || !MI355_NT_LOADS ? 0 : 2 /* nt */);
            c[d] = make_uint4(v.x, v.y, v.z, v.w);
            if (PAIR) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(a.prev + (size_t)t * a.stride, a.n), voff, 0, 0);
                p[d] = make_uint4(w.x, w.y, w.z, w.w);
            }
        }
#if MI355_K1_PIN_LOADS
        __builtin_amdgcn_sched_barrier(0);   // the loads stay in front of the arithmetic of the other group
#endif
    }
};

// One frame of one full tile: compare, feed back, append the candidates' records at log position `run`.
__device__ __forceinline__ uint64_t pack_step2(const uint4 c, uint4 &s, ThrConst tc, __amdgpu_buffer_rsrc_t rec,
                                               uint32_t run, uint32_t tile, uint32_t ntiles, uint32_t &cnt4) {
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
    uint32_t sw[4] = {s.x, s.y, s.z, s.w};
    uint32_t dm[4], sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t x;
        const uint32_t fh = dword_flags(cw[k], sw[k], tc, x);
        sel[k] = perm_select(fh);
        const uint32_t d = bytes_sub_from_x(cw[k], sw[k], x);
        dm[k] = __builtin_amdgcn_perm(d, 0u, sel[k]);
        sw[k] = __builtin_amdgcn_perm(cw[k], sw[k], sel[k]);     // negative feedback (kernels.cu:316-331)
    }
    s = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    cnt4 = __builtin_amdgcn_sad_u8(sel[0] + sel[1] + sel[2] + sel[3], 0u, 0u);
    const bool cand = ((dm[0] | dm[1]) | (dm[2] | dm[3])) != 0u;
    const uint64_t mask = __ballot(cand);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    // byte offset of log position run + rank (rec_index): the chunk of `run` is wave-uniform, a record that
    // falls into the next chunk lies (ntiles - 1) KiB further on
    const uint32_t sbase = (__umul24(run >> 6, ntiles) + tile) * 1024u;             // scalar
    const uint32_t q = (run & 63u) + rank;
    const uint32_t off = sbase + q * 16u + (q >= 64u ? (ntiles - 1u) * 1024u : 0u);
#if MI355_ABLATE == 1 || MI355_ABLATE == 2
    asm volatile("" ::"v"(dm[0]), "v"(dm[1]), "v"(dm[2]), "v"(dm[3]), "v"(off));
#else
    const u32x4 v = {dm[0], dm[1], dm[2], dm[3]};
    __builtin_amdgcn_raw_buffer_store_b128(v, rec, cand ? off : kOOB, 0, 0);
#endif
    return mask;
}

template <bool PAIR>
__device__ __forceinline__ void pack_group2(const PackArgs &a, const Group2<PAIR> &g, int t0, uint4 &st, uint32_t &run,
                                            uint32_t tile, ThrConst tc, int lane, __amdgpu_buffer_rsrc_t rec,
                                            __amdgpu_buffer_rsrc_t metab) {
    uint4 meta = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < kPrefetch; d += 2) {
        uint32_t c0, c1;
        if (PAIR) st = g.p[d];
        const uint32_t run0 = run;
        const uint64_t m0 = pack_step2(g.c[d], st, tc, rec, run, tile, a.ntiles, c0);
        run += (uint32_t)__builtin_popcountll(m0);
        const uint32_t run1 = run;
        if (PAIR) st = g.p[d + 1];
        const uint64_t m1 = pack_step2(g.c[d + 1], st, tc, rec, run, tile, a.ntiles, c1);
        run += (uint32_t)__builtin_popcountll(m1);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(wave_inclusive_scan((int)(c0 | (c1 << 16))), 63);
        const uint32_t n0 = ((tot & 0xffffu) - 64u * 24u) >> 2, n1 = ((tot >> 16) - 64u * 24u) >> 2;
        write_lane(meta.x, (uint32_t)m0, d);
        write_lane(meta.y, (uint32_t)(m0 >> 32), d);
        write_lane(meta.z, n0, d);
        write_lane(meta.w, run0, d);
        write_lane(meta.x, (uint32_t)m1, d + 1);
        write_lane(meta.y, (uint32_t)(m1 >> 32), d + 1);
        write_lane(meta.z, n1, d + 1);
        write_lane(meta.w, run1, d + 1);
    }
#if MI355_ABLATE != 2
    const uint32_t moff = (__umul24((uint32_t)t0 + (uint32_t)lane, a.ntiles) + tile) * 16u;
    const u32x4 mv = {meta.x, meta.y, meta.z, meta.w};
    __builtin_amdgcn_raw_buffer_store_b128(mv, metab, lane < kPrefetch ? moff : kOOB, 0, 0);
#endif
}

template <bool PAIR, bool FAST>
__device__ __forceinline__ void pack_tile(const PackArgs &a, uint32_t tile, uint32_t byte_off,
                                          int valid, int lane) {
    const int T = a.nframes;
    const ThrConst tc{(127u - (uint32_t)a.thr) * 0x01010101u, (uint32_t)a.thr * 0x01010101u};

    uint4 st = make_uint4(0, 0, 0, 0);
    if (!PAIR) st = load16<FAST>(a.state + byte_off, valid);
    uint32_t run = 0;  // records this tile has appended to its log so far
    int t0 = 0;

#if MI355_K1V == 2
    if (FAST) {
        // complete groups, two per iteration (the register groups swap roles), exact waits (see above)
        const __amdgpu_buffer_rsrc_t rec = make_rsrc(a.rec, a.rec_bytes);
        const __amdgpu_buffer_rsrc_t metab = make_rsrc(a.meta, a.meta_bytes);
        Group2<PAIR> ga, gb;
        // two groups per round; the first round is peeled so that the loop is entered with exactly the operations
        // in flight that its back edge carries (the compiler merges the wait counts of both entries to the smaller)
        auto round = [&]() __attribute__((always_inline)) {
            gb.load(a, byte_off, t0 + kPrefetch);
            pack_group2<PAIR>(a, ga, t0, st, run, tile, tc, lane, rec, metab);
            ga.load(a, byte_off, t0 + 2 * kPrefetch);
            pack_group2<PAIR>(a, gb, t0 + kPrefetch, st, run, tile, tc, lane, rec, metab);
            t0 += 2 * kPrefetch;
        };
        if (T >= 2 * kPrefetch) {
            ga.load(a, byte_off, 0);
            round();
            while (t0 + 2 * kPrefetch <= T) round();
        }
    }
#endif
    // the general form: the (up to 2 * kPrefetch - 1) frames the steady state leaves, batches shorter than
    // that, and ragged tiles
    if (t0 < T) {
        Group<PAIR, FAST> ga, gb;
        ga.load(a, byte_off, t0, valid);
        for (;;) {
            gb.load(a, byte_off, t0 + kPrefetch, valid);
            pack_group<PAIR, FAST>(a, ga, t0, st, run, tile, tc, lane);
            t0 += kPrefetch;
            if (t0 >= T) break;
            ga.load(a, byte_off, t0 + kPrefetch, valid);
            pack_group<PAIR, FAST>(a, gb, t0, st, run, tile, tc, lane);
            t0 += kPrefetch;
            if (t0 >= T) break;
        }
    }

    if (!PAIR) {
        if (FAST) *reinterpret_cast<uint4 *>(a.state + byte_off) = st;
        else store16_bytes(a.state + byte_off, st, valid);
    }
}

template <bool PAIR, bool ALIGNED>
__global__ __launch_bounds__(256) void k_diff_pack(const PackArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (tile >= a.ntiles) return;  // wave-uniform
    const uint32_t tile_off = tile * kTileBytes;
    const uint32_t byte_off = tile_off + (uint32_t)lane * 16u;
    // wave-uniform choice: every lane of a full, aligned tile takes the vector path
    if (ALIGNED && tile_off + kTileBytes <= a.n) {
        pack_tile<PAIR, true>(a, tile, byte_off, 16, lane);
    } else {
        const int valid = byte_off < a.n ? (int)min(16u, a.n - byte_off) : 0;
        pack_tile<PAIR, false>(a, tile, byte_off, valid, lane);
    }
}

hipError_t launch_diff_pack(const PackArgs &a, bool pair, bool aligned, hipStream_t s) {
    const dim3 block(64 * kWavesPerBlock);
    const dim3 grid((a.ntiles + kWavesPerBlock - 1) / kWavesPerBlock);
    if (pair) {
        if (aligned) hipLaunchKernelGGL((k_diff_pack<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_diff_pack<true, false>), grid, block, 0, s, a);
    } else {
        if (aligned) hipLaunchKernelGGL((k_diff_pack<false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_diff_pack<false, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

// ---- scans ----------------------------------------------------------------------------------------
// Block-wide exclusive scan of one value per thread (NW waves).
template <int NW>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds /*NW+1*/,
                                                         uint32_t &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t x = lds[w];
            lds[w] = acc;
            acc += x;
        }
        lds[NW] = acc;
    }
    __syncthreads();
    const uint32_t r = lds[wave] + incl - v;
    block_total = lds[NW];
    __syncthreads();
    return r;
}

// A *group* is kXTiles consecutive tiles (four k_expand waves); the expander needs, per frame,
// the bytes in the groups before its own and rebuilds everything finer from the meta words.
constexpr uint32_t kXTiles = 64;          // = one wave of k_scan_groups per group
constexpr uint32_t kScanChunk = 1024;     // groups scanned per pass of k_scan_groups

// The one place where the library orders two agent-scope accesses without a release fence: `*slot = value` must be
// visible to whoever sees the ticket this call takes.  The store is an agent-scope (write-through) store and the
// lane waits for its completion (s_waitcnt vmcnt(0): the write has reached the level all XCDs share) before it
// issues the relaxed ticket increment.  A release fence at agent scope would do the same and also write back this
// XCD's whole L2, which at this point is full of k_diff_pack's fresh log lines: 19 us instead of 12 us per launch
// (profiles/README.md).  This leans on gfx950's memory pipeline, not on the HIP memory model; the reader uses
// agent-scope loads.  Guard: tests/soak.py (15 000 random batches against the oracle, clean).
__device__ __forceinline__ uint32_t publish_then_take_ticket(uint32_t *slot, uint32_t value, uint32_t *ticket) {
    __hip_atomic_store(slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid = T, block = 256: groff[t][g] = flagged bytes of frame t in the groups before g, totals[t] = all
// of them.  A wave loads the 64 byte counts of a group with one coalesced instruction and reduces them
// with DPP; the (at most kScanChunk) group sums are scanned in LDS.
// The workgroup that finishes last (ticket counter, reset for the next launch) also scans the frame totals
// into offsets[0..T]: one launch and one dependent round trip less than a separate kernel.
__global__ __launch_bounds__(256) void k_scan_groups(const uint4 *meta, uint32_t *groff, uint32_t *totals,
                                                     uint32_t ntiles, uint32_t ngroups, uint32_t *ticket,
                                                     uint32_t *offsets) {
    static_assert(kXTiles == 64, "one wave reduces one group");
    __shared__ uint32_t s_sum[kScanChunk];
    __shared__ uint32_t s_scan[5];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row = (size_t)blockIdx.x * ntiles;
    uint32_t *out = groff + (size_t)blockIdx.x * ngroups;
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < ngroups; g0 += kScanChunk) {
        const uint32_t gn = min(kScanChunk, ngroups - g0);
        for (uint32_t g = wave * 4; g < gn; g += 16) {   // four groups per wave and round: four loads in flight
            uint32_t z[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t tile = (g0 + g + k) * kXTiles + lane;
                z[k] = (g + k < gn && tile < ntiles) ? meta[row + tile].z : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t incl = (uint32_t)wave_inclusive_scan((int)z[k]);
                if (lane == 63 && g + k < gn) s_sum[g + k] = incl;
            }
        }
        __syncthreads();
        uint32_t v[4], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t idx = threadIdx.x * 4 + i;
            v[i] = idx < gn ? s_sum[idx] : 0u;
            sum += v[i];
        }
        uint32_t total;
        uint32_t acc = carry + block_exclusive_scan<4>(sum, s_scan, total);   // ends with a barrier
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t idx = threadIdx.x * 4 + i;
            if (idx < gn) out[g0 + idx] = acc;
            acc += v[i];
        }
        carry += total;
    }
    // totals and the ticket are agent-scope atomics: the workgroups run on different XCDs, whose L2s are
    // not coherent for plain accesses (publish_then_take_ticket says how the two are ordered).
    __shared__ uint32_t s_is_last;
    if (threadIdx.x == 0) s_is_last = publish_then_take_ticket(&totals[blockIdx.x], carry, ticket) == gridDim.x - 1;
    __syncthreads();
    if (!s_is_last) return;
    const uint32_t nframes = gridDim.x;
    carry = 0;
    for (uint32_t base = 0; base < nframes; base += 256) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < nframes
            ? __hip_atomic_load(&totals[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan<4>(v, s_scan, total);
        if (t < nframes) offsets[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        offsets[nframes] = carry;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

uint32_t expand_groups(uint32_t ntiles) { return (ntiles + kXTiles - 1) / kXTiles; }

hipError_t launch_scan(const uint4 *meta, uint32_t *groff, uint32_t *totals, uint32_t ntiles,
                       int nframes, uint32_t *offsets, uint32_t *ticket, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_groups, dim3(nframes), dim3(256), 0, s, meta, groff, totals, ntiles,
                       expand_groups(ntiles), ticket, offsets);
    return hipGetLastError();
}

// ---- expand: records -> packed frame-major (xs, diff) -------------------------------------------------
// grid = (ceil(W/16) rounded up to a multiple of 8, T), block = 64: one wave owns 16 consecutive tiles of
// ONE frame, i.e. one contiguous range of that frame's output.  The wave loads the meta words of its whole
// 64-tile group (the scan kernel gives the bytes in front of the group, the wave adds those in front of
// its own 16 tiles) and keeps the per-tile facts in registers (readlane / ds_bpermute).  For every
// candidate lane of its tiles it writes (tile, lane) at the record's rank into a small LDS table (ballot +
// mbcnt), so that "record r of the wave" is found with one LDS read; 64 records are loaded per round,
// their 16-bit maps of nonzero (= flagged) bytes come from v_dot4_u32_u8, the entry offsets from one DPP
// scan; entries are staged in LDS in output order and leave with coalesced stores.  No barriers.
// Measured (profiles/README.md): the kernel is bound by the log's round trip through the memory system
// (without its record reads it takes 0.13 ms instead of 0.23 ms per 256-frame batch -- and the NEXT
// k_diff_pack then takes 0.09 ms longer, because its record stores no longer find their lines cached);
// instruction count, occupancy, barriers and XCD placement were each varied without effect.
#ifndef MI355_XLIGHT
#define MI355_XLIGHT 4
#endif
constexpr uint32_t kXLight = MI355_XLIGHT;   // records with more flagged bytes than this are "heavy"
constexpr int kXHeavyMax = 12;               // more heavy records than this in a wave: everybody walks

// WIRE: the entries leave in the sender's byte stream instead (server/src/threads.cpp:227-229): frame t
// is {u32 n, i32 xs[n], u8 diff[n]} at byte 4t + 5*offsets[t] of a.wire, so index and payload sections
// start at arbitrary byte addresses (gfx950 global stores need no alignment).
__device__ __forceinline__ void store_u32_unaligned(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// Entries [first, first + count) of a workgroup of NT threads, staged in LDS at s_xs/s_df[0..count), leave with coalesced
// stores: one dword per index, and the differences as whole dwords too (byte stores only for the up to
// three bytes before and after the dword-aligned body of the destination).
// Hand-off between lanes of ONE wave through LDS (k_expand is a single-wave workgroup): the DS operations of a
// wave execute in order, so this costs nothing in hardware; it keeps the compiler from moving LDS accesses
// across the hand-off.
__device__ __forceinline__ void lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool WIRE, uint32_t NT>
__device__ __forceinline__ void flush_entries(const ExpandArgs &a, const uint16_t *s_xs, const uint8_t *s_df,
                                              uint32_t first, uint32_t count, uint32_t xs0, uint32_t dst0,
                                              uint8_t *w_xs, uint8_t *w_df, size_t w_room) {
    uint8_t *xsp, *dfp;
    uint32_t n;
    if (WIRE) {
        n = w_room ? count : 0u;
        xsp = w_xs + 4 * (size_t)first;
        dfp = w_df + first;
    } else {
        const size_t d = (size_t)dst0 + first;      // entries beyond the capacity are dropped
        n = d >= a.capacity ? 0u : (uint32_t)(a.capacity - d < count ? a.capacity - d : count);
        xsp = (uint8_t *)(a.out_xs + d);
        dfp = a.out_diff + d;
    }
    for (uint32_t i = threadIdx.x; i < n; i += NT) store_u32_unaligned(xsp + 4 * (size_t)i, xs0 + s_xs[i]);
    const uint32_t lead = (4u - (uint32_t)((uintptr_t)dfp & 3u)) & 3u;
    const uint32_t head = lead < n ? lead : n;
    const uint32_t body = (n - head) >> 2;
    if (threadIdx.x < head) dfp[threadIdx.x] = s_df[threadIdx.x];
    for (uint32_t k = threadIdx.x; k < body; k += NT) {
        const uint8_t *q = s_df + head + 4 * k;
        const uint32_t v = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        *reinterpret_cast<uint32_t *>(dfp + head + 4 * (size_t)k) = v;
    }
    const uint32_t tail = head + 4 * body + threadIdx.x;
    if (tail < n) dfp[tail] = s_df[tail];
}

constexpr uint32_t kWTiles = 16;       // tiles per single-wave workgroup
constexpr uint32_t kWStage = 1024;     // entries staged per wave = the most a round of 64 records can hold (5 KB of LDS per wave in all)

template <bool WIRE>
__global__ __launch_bounds__(64) void k_expand(const ExpandArgs a) {
    __shared__ uint16_t s_src[kWTiles * 64];
    __shared__ uint16_t s_xs[kWStage];
    __shared__ uint8_t s_df[kWStage];
    const int lane = threadIdx.x;
    const int t = blockIdx.y;
    const uint32_t sub = blockIdx.x, group = sub >> 2, q = sub & 3u;
    const uint32_t tile0 = sub * kWTiles, gtile0 = group * kXTiles;
    const uint32_t ngroups = (a.ntiles + kXTiles - 1) / kXTiles;
    if (tile0 >= a.ntiles) return;   // grid.x is padded to a multiple of 8 (see launch_expand)
    const size_t row = (size_t)t * a.ntiles;
    const uint32_t off_t = a.offsets[t];
    const uint32_t goff = a.groff[(size_t)t * ngroups + group];
    uint32_t n_t = 0;
    size_t head = 0;
    if (WIRE) {
        n_t = a.offsets[t + 1] - off_t;
        head = 4 * (size_t)t + 5 * (size_t)off_t;
        if (sub == 0 && lane == 0 && head + 4 <= a.capacity) store_u32_unaligned(a.wire + head, n_t);
    }
    // lane L <-> tile gtile0 + L of the group; lanes 16q .. 16q+15 are this wave's tiles
    uint4 m = make_uint4(0, 0, 0, 0);
    if (gtile0 + (uint32_t)lane < a.ntiles) m = a.meta[row + gtile0 + (uint32_t)lane];
    const uint64_t mask = (uint64_t)m.x | ((uint64_t)m.y << 32);
    const bool mine = ((uint32_t)lane >> 4) == q;
    const uint32_t nr = mine ? (uint32_t)__builtin_popcountll(mask) : 0u;
    const uint32_t rincl = (uint32_t)wave_inclusive_scan((int)nr);
    const uint32_t bincl = (uint32_t)wave_inclusive_scan((int)m.z);
    const uint32_t nrec = (uint32_t)__builtin_amdgcn_readlane((int)rincl, 63);
    if (nrec == 0) return;
    const uint32_t before = q ? (uint32_t)__builtin_amdgcn_readlane((int)bincl, (int)(16u * q - 1u)) : 0u;
    const uint32_t dst0 = off_t + goff + before;   // < 2^32: the batch total is below 2^32
    const uint32_t rexcl = rincl - nr;             // records of this wave before the lane's tile
    const uint32_t rbase = m.w - rexcl;            // + record index in the wave = log position of the record
    uint8_t *w_xs = nullptr, *w_df = nullptr;
    size_t w_room = 0;
    if (WIRE) {
        const size_t end = head + 4 + 5 * (size_t)n_t;
        const uint32_t seg = dst0 - off_t;
        w_xs = a.wire + head + 4 + 4 * (size_t)seg;
        w_df = a.wire + head + 4 + 4 * (size_t)n_t + seg;
        w_room = end <= a.capacity ? (size_t)n_t : 0;
    }
    // where each record of the wave comes from: candidate lanes write (tile, lane) at the record's rank
#pragma unroll 4
    for (uint32_t i = 0; i < kWTiles; i++) {
        const int L = (int)(16u * q + i);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mask, L);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mask >> 32), L);
        const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)rexcl, L);
        const uint64_t mk = (uint64_t)lo | ((uint64_t)hi << 32);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
        if ((mk >> lane) & 1) s_src[rx + rank] = (uint16_t)((i << 6) | (uint32_t)lane);
    }
    lds_handoff();   // s_src is read by other lanes than the ones that wrote it
    const uint32_t xs_base = tile0 * kTileBytes;
    uint32_t carry = 0, flushed = 0;   // entries emitted / already stored
    for (uint32_t base = 0; base < nrec; base += 64) {
        const uint32_t r = base + (uint32_t)lane;
        const uint32_t src = r < nrec ? s_src[r] : 0u;
        const uint32_t sgm = src >> 6;
        // every lane takes part in the permute; the value comes from the lane that holds tile sgm of this wave
        const uint32_t rb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((16u * q + sgm) * 4u), (int)rbase);
        const uint32_t src16 = src * 16u;
        uint4 rec = make_uint4(0, 0, 0, 0);
        if (r < nrec) rec = a.rec[rec_index(r + rb, tile0 + sgm, a.ntiles)];
        const uint32_t m16 = record_map16(rec);
        const uint32_t cnt = (uint32_t)__builtin_popcount(m16);
        const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
        const uint32_t round_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (carry - flushed + round_total > kWStage) {   // wave-uniform: make room
            lds_handoff();
            flush_entries<WIRE, 64>(a, s_xs, s_df, flushed, carry - flushed, xs_base, dst0, w_xs, w_df, w_room);
            lds_handoff();   // the stage is rewritten from its start
            flushed = carry;
        }
        uint32_t e = carry - flushed + incl - cnt;        // index in the LDS stage
        carry += round_total;
        uint32_t mm = m16;
        const uint64_t heavy = __ballot(cnt > kXLight);
        const bool coop = __builtin_popcountll(heavy) <= kXHeavyMax;
        const uint32_t e0 = e;
        if (coop && cnt > kXLight) mm = 0;
        while (mm) {
            const int b = __builtin_ctz(mm);
            mm &= mm - 1;
            const uint32_t dw = b < 8 ? (b < 4 ? rec.x : rec.y) : (b < 12 ? rec.z : rec.w);
            s_xs[e] = (uint16_t)(src16 + (uint32_t)b);                       // kernels.cu:315
            s_df[e] = (uint8_t)(dw >> (8 * (b & 3)));                        // kernels.cu:314
            ++e;
        }
        if (coop) {
            const uint32_t b = (uint32_t)lane & 15u;
            for (uint64_t h = heavy; h; h &= h - 1) {
                const int hl = __builtin_ctzll(h);
                const uint32_t r0 = __builtin_amdgcn_readlane(rec.x, hl);
                const uint32_t r1 = __builtin_amdgcn_readlane(rec.y, hl);
                const uint32_t r2 = __builtin_amdgcn_readlane(rec.z, hl);
                const uint32_t r3 = __builtin_amdgcn_readlane(rec.w, hl);
                const uint32_t hm = __builtin_amdgcn_readlane(m16, hl);
                const uint32_t ee = __builtin_amdgcn_readlane(e0, hl);
                const uint32_t sb = __builtin_amdgcn_readlane(src16, hl);
                if (lane < 16 && ((hm >> b) & 1u)) {
                    const uint32_t pos = ee + (uint32_t)__builtin_popcount(hm & ((1u << b) - 1u));
                    const uint32_t dw = b < 8 ? (b < 4 ? r0 : r1) : (b < 12 ? r2 : r3);
                    s_xs[pos] = (uint16_t)(sb + b);
                    s_df[pos] = (uint8_t)(dw >> (8 * (b & 3)));
                }
            }
        }
    }
    lds_handoff();
    flush_entries<WIRE, 64>(a, s_xs, s_df, flushed, carry - flushed, xs_base, dst0, w_xs, w_df, w_room);
}

hipError_t launch_expand(const ExpandArgs &a, int nframes, hipStream_t s) {
    // Workgroups go to the 8 XCDs round-robin by linear id: with grid.x a multiple of 8 the workgroups of one
    // tile range land on the same XCD for every frame (the padding workgroups return at once).
    const uint32_t gx = (a.ntiles + kWTiles - 1) / kWTiles;
    const dim3 grid((gx + 7u) / 8u * 8u, nframes);
    if (a.wire)
        hipLaunchKernelGGL(k_expand<true>, grid, dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(k_expand<false>, grid, dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace mi355
