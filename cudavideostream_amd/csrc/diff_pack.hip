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
//                   flags  10 ops  exact 9-bit compare of 4 bytes (dword_flags)
//                   select  2 ops  v_perm selector from the flags
//                   state   1 op   v_perm(cur, state)          negative feedback
//                   diff    3 ops  per-byte (cur - state), zeroed where un-flagged
//                   map     1.5 ops (v_dot4 of the flag bytes: 16-bit map of the lane's flagged bytes)
//                 A lane with at least one flagged byte is a *candidate*: ranked with one ballot + mbcnt
//                 it appends a 4-byte code {map, the byte's difference, lane} to the tile's code log; a lane
//                 with two or more flagged bytes also appends its 16 masked difference bytes to the record
//                 log (see "the log" below).  Per (frame, tile) one 16-byte meta word.
//   k_scan_groups: per frame, flagged bytes before every range of 16 tiles; its last workgroup scans the
//               frame totals into offsets[0..T].
//   k_expand    : one wave per (frame, 16 tiles): turns codes and records into the caller's packed,
//                 frame-major, ascending (xs, diff) arrays -- or the socket's byte stream -- through an
//                 LDS stage and coalesced stores.
// No spin waits; the only inter-workgroup communication is the completion ticket of k_scan_groups;
// results are independent of dispatch order.
#include "pack_common.h"

namespace mi355 {

// Ablation builds (tools/ab_build.sh, never shipped): 1 = no log stores, 2 = also no meta stores,
// 3 = loads + state fold only (memory floor of the stream loop).  0 = the product.
#ifndef MI355_ABLATE
#define MI355_ABLATE 0
#endif

#ifndef MI355_K1_PREFETCH
#define MI355_K1_PREFETCH 4
#endif
// Frames per register group (two groups per wave).  Stream mode: 4 (8 x 1 KiB in flight per wave, 58 VGPRs).  Pair
// mode holds two operands per frame: with 4 it needed 89 VGPRs = 5 waves per SIMD, and the 6076 waves of a 1080p frame
// no longer fitted the chip at once (5120 places): a sixth of them ran as a second generation, alone.  With 2 the
// pair kernel fits 6 waves per SIMD like the stream kernel (the depth of the prefetch was measured not to matter).
template <bool PAIR>
struct PrefetchOf { static constexpr int value = PAIR ? (MI355_K1_PREFETCH > 2 ? 2 : MI355_K1_PREFETCH) : MI355_K1_PREFETCH; };
static_assert(MI355_K1_PREFETCH % 2 == 0, "frames are processed in pairs");

// ---- the log (round 3) ------------------------------------------------------------------------------
// What k_diff_pack leaves for k_expand, per (frame, tile):
//   * one 4-byte CODE per candidate lane (a lane with >= 1 flagged byte), in lane order:
//         bits  0..15  map of the lane's flagged bytes
//         bits 16..23  ONE flagged byte: its difference;  more: the lane's rank among the tile's
//                      multi-byte lanes of this frame (where its record is)
//         bits 24..29  the lane
//   * one 16-byte RECORD (the 16 masked difference bytes) per lane with >= 2 flagged bytes,
//   * one 16-byte meta word {byte offset of the frame's first code, of its first record, flagged bytes,
//     candidates | multi-byte lanes << 16}.
// On webcam-like input (isolated bytes) 84 % of the candidates carry one byte: the log shrinks from 16 to
// 4 + 0.16 * 16 = 6.6 bytes per candidate, and the expander neither rebuilds the byte maps nor searches for the
// lane a record came from.  Both logs are chunk-interleaved: chunk k of tile `tile` is the KiB at
// (k * ntiles + tile) * 1024 (every tile appends sequentially; all tiles' current chunks are neighbours in
// memory); a chunk holds 256 codes or 64 records, and a frame's units never straddle two chunks (LogPos).
// Stores are raw buffer stores that are always issued: a lane with nothing to store carries an offset beyond
// the descriptor's range and the hardware drops it (no branch around the store, no traffic).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kRsrcWord3 = 0x00020000u;   // raw buffer, 32-bit data format (gfx9 family)
constexpr uint32_t kOOB = 0xFFFFFFFFu;         // beyond every descriptor's num_records: the lane's store is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, (int)kRsrcWord3);
}

struct LogOut {
    __amdgpu_buffer_rsrc_t codes, recs, meta;
};

// One frame of one tile, arithmetic only: compare, feed back.  dm = the 16 masked difference bytes,
// m16 = map of the flagged bytes.
template <bool HIGH>
__device__ __forceinline__ void compare_step(const uint4 c, uint4 &s, ThrConst tc, uint32_t (&dm)[4], uint32_t &m16) {
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
    uint32_t sw[4] = {s.x, s.y, s.z, s.w};
    uint32_t fh[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
#if MI355_ABLATE == 3
        sw[k] ^= cw[k]; dm[k] = 0; fh[k] = 0;
#else
        uint32_t x;
        fh[k] = dword_flags<HIGH>(cw[k], sw[k], tc, x);
        // 0xFF in every flagged byte: a v_perm selector byte of 0x80 yields the constant 0xFF, one of 0x00
        // byte 0 of the second operand (0)
        const uint32_t mask = __builtin_amdgcn_perm(0u, 0u, fh[k]);
        dm[k] = bytes_sub_from_x(cw[k], sw[k], x) & mask;     // diff where flagged, 0 elsewhere
        // negative feedback (kernels.cu:316-331): flagged bytes take the current value, the others
        // keep the previous one -> the state is the frame the client reconstructs
        sw[k] = bitop3<(TA & TC) | (TB & ~TC)>(cw[k], sw[k], mask);
#endif
    }
    s = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    // flags are 0x80 per flagged byte: two v_dot4 chains weigh them into 128 * (map of 8 bytes)
    const uint32_t lo = __builtin_amdgcn_udot4(fh[1], 0x80402010u, __builtin_amdgcn_udot4(fh[0], 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(fh[3], 0x80402010u, __builtin_amdgcn_udot4(fh[2], 0x08040201u, 0u, false), false);
    m16 = (lo + (hi << 8)) >> 7;
}

// Where the tile's next code / record goes (wave-uniform, SGPRs): byte offset into the log and units left in
// the current KiB chunk.  A frame's codes (records) never straddle a chunk: if they do not fit, the rest of the
// chunk is skipped (the next chunk of the same tile lies `jump` = (ntiles - 1) KiB behind the end of this one),
// so a frame's units are contiguous and k_expand addresses them with the byte offset kept in the meta word.
struct LogPos {
    uint32_t ptrC, roomC, ptrM, roomM;
};

// Appends the frame's codes and records; pc / pm = byte offsets of its first code / record; returns
// candidates | multi-byte lanes << 16.
__device__ __forceinline__ uint32_t emit_step(const uint32_t (&dm)[4], uint32_t m16, const LogOut &lg, LogPos &lp,
                                              uint32_t jump, uint32_t lane24, uint32_t &pc, uint32_t &pm) {
    const bool cand = m16 != 0u;
    const bool multi = (m16 & (m16 - 1u)) != 0u;
    const uint64_t bc = __ballot(cand), bm = __ballot(multi);
    const uint32_t rankC = __builtin_amdgcn_mbcnt_hi((uint32_t)(bc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bc, 0u));
    const uint32_t rankM = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
    const uint32_t nc = (uint32_t)__builtin_popcountll(bc), nm = (uint32_t)__builtin_popcountll(bm);
    pc = lp.ptrC;
    pm = lp.ptrM;
    if (nc == 0u) return 0u;   // wave-uniform: a still tile appends nothing (and issues no store)
    if (nc > lp.roomC) { lp.ptrC += lp.roomC * 4u + jump; lp.roomC = 256u; }
    if (nm > lp.roomM) { lp.ptrM += lp.roomM * 16u + jump; lp.roomM = 64u; }
    pc = lp.ptrC;
    pm = lp.ptrM;
    // a lane with one flagged byte: the byte sum of its masked differences IS that byte
    const uint32_t one = __builtin_amdgcn_sad_u8((dm[0] | dm[1]) | (dm[2] | dm[3]), 0u, 0u);
    const uint32_t code = m16 | ((multi ? rankM : one) << 16) | lane24;
#if MI355_ABLATE == 0
    __builtin_amdgcn_raw_buffer_store_b32(code, lg.codes, cand ? pc + rankC * 4u : kOOB, 0, 0);
    const u32x4 v = {dm[0], dm[1], dm[2], dm[3]};
    __builtin_amdgcn_raw_buffer_store_b128(v, lg.recs, multi ? pm + rankM * 16u : kOOB, 0, 0);
#else
    asm volatile("" ::"v"(dm[0]), "v"(dm[1]), "v"(dm[2]), "v"(dm[3]), "v"(rankC), "v"(rankM), "v"(code));
#endif
    lp.ptrC += nc * 4u;
    lp.roomC -= nc;
    lp.ptrM += nm * 16u;
    lp.roomM -= nm;
    return nc | (nm << 16);
}

// A group = kPrefetch consecutive frames of one tile held in registers.  Loads are always issued
// (frame index clamped to T-1): on gfx950 loads and stores share one in-order vmcnt, and a conditional
// load makes the compiler fall back to s_waitcnt vmcnt(0) -- i.e. no prefetch.
template <bool PAIR, bool FAST>
struct Group {
    static constexpr int kPrefetch = PrefetchOf<PAIR>::value;
    uint4 c[kPrefetch];
    uint4 p[kPrefetch];

    // The frame bases are wave-uniform (SGPRs) and the lane's byte offset is a 32-bit VGPR: the loads use the
    // SGPR-base + VGPR-offset form.  `cur0` / `prev0` point at frame t0; frames beyond the batch repeat the last one.
    __device__ __forceinline__ void load(const PackArgs &a, uint32_t byte_off, int t0, int valid, const uint8_t *cur0,
                                         const uint8_t *prev0, const uint8_t *cur_last, const uint8_t *prev_last) {
        const int last = a.nframes - 1;
#pragma unroll
        for (int d = 0; d < kPrefetch; d++) {
            const bool in = t0 + d <= last;
            const uint8_t *cb = uniform_ptr(in ? cur0 + (size_t)d * a.stride : cur_last);
            c[d] = load16<FAST, !PAIR>(cb + byte_off, valid);   // stream frames: read once
            if (PAIR) {
                const uint8_t *pb = uniform_ptr(in ? prev0 + (size_t)d * a.stride : prev_last);
                p[d] = load16<FAST>(pb + byte_off, valid);
            }
        }
    }
};

template <bool PAIR, bool FAST, bool HIGH>
__device__ __forceinline__ void pack_group(const PackArgs &a, const Group<PAIR, FAST> &g, int t0, uint4 &st,
                                           LogPos &lp, uint32_t tile, ThrConst tc, int lane, const LogOut &lg) {
    // The group's kPrefetch meta words are assembled in lanes 0..kPrefetch-1 and leave with ONE store.
    constexpr int kPrefetch = PrefetchOf<PAIR>::value;
    uint4 meta = make_uint4(0, 0, 0, 0);
    const uint32_t lane24 = (uint32_t)lane << 24;
    const uint32_t jump = (a.ntiles - 1u) * 1024u;
#pragma unroll
    for (int d = 0; d < kPrefetch; d += 2) {
        const int t = t0 + d;
        if (t >= a.nframes) break;  // wave-uniform
        const bool two = t + 1 < a.nframes;
        uint32_t dm0[4], m0, m1 = 0, pc0, pm0, pc1 = 0, pm1 = 0;
        if (PAIR) st = g.p[d];
        compare_step<HIGH>(g.c[d], st, tc, dm0, m0);
        const uint32_t c0 = emit_step(dm0, m0, lg, lp, jump, lane24, pc0, pm0);
        uint32_t c1 = 0;
        if (two) {
            if (PAIR) st = g.p[d + 1];
            uint32_t dm1[4];
            compare_step<HIGH>(g.c[d + 1], st, tc, dm1, m1);
            c1 = emit_step(dm1, m1, lg, lp, jump, lane24, pc1, pm1);
        }
        // flagged bytes of the two frames: one register (16-bit fields, a tile holds at most 1024) and one
        // DPP reduction for both
        const uint32_t both = (uint32_t)__builtin_popcount(m0) | ((uint32_t)__builtin_popcount(m1) << 16);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane(wave_inclusive_scan((int)both), 63);
        // all values are wave-uniform: v_writelane drops them into lanes d and d+1
        write_lane(meta.x, pc0, d);
        write_lane(meta.y, pm0, d);
        write_lane(meta.z, tot & 0xffffu, d);
        write_lane(meta.w, c0, d);
        write_lane(meta.x, pc1, d + 1);
        write_lane(meta.y, pm1, d + 1);
        write_lane(meta.z, tot >> 16, d + 1);
        write_lane(meta.w, c1, d + 1);
    }
#if MI355_ABLATE != 2 && MI355_ABLATE != 3
    const uint32_t moff = (__umul24((uint32_t)t0 + (uint32_t)lane, a.ntiles) + tile) * 16u;
    const u32x4 mv = {meta.x, meta.y, meta.z, meta.w};
    __builtin_amdgcn_raw_buffer_store_b128(mv, lg.meta, (lane < kPrefetch && t0 + lane < a.nframes) ? moff : kOOB, 0, 0);
#endif
}

template <bool PAIR, bool FAST, bool HIGH>
__device__ __forceinline__ void pack_tile(const PackArgs &a, uint32_t tile, uint32_t byte_off,
                                          int valid, int lane) {
    const int T = a.nframes;
    const ThrConst tc = make_thr((uint32_t)a.thr);
    const LogOut lg{make_rsrc(a.codes, a.codes_bytes), make_rsrc(a.rec, a.rec_bytes), make_rsrc(a.meta, a.meta_bytes)};

    uint4 st = make_uint4(0, 0, 0, 0);
    if (!PAIR) st = load16<FAST>(a.state + byte_off, valid);

    // Two register groups: while one is processed (its stores are issued), the other's loads are in
    // flight.
    Group<PAIR, FAST> ga, gb;
    LogPos lp{tile * 1024u, 256u, tile * 1024u, 64u};   // codes / records this tile has appended to its logs so far
    constexpr int kPrefetch = PrefetchOf<PAIR>::value;
    const size_t gstep = (size_t)kPrefetch * a.stride;
    const uint8_t *cur_last = a.cur + (size_t)(T - 1) * a.stride, *prev_last = PAIR ? a.prev + (size_t)(T - 1) * a.stride : nullptr;
    const uint8_t *cp = a.cur, *pp = a.prev;   // frame t0 + kPrefetch, the next group to request
    ga.load(a, byte_off, 0, valid, cp, pp, cur_last, prev_last);
    for (int t0 = 0;;) {
        cp += gstep; if (PAIR) pp += gstep;
        gb.load(a, byte_off, t0 + kPrefetch, valid, cp, pp, cur_last, prev_last);
        pack_group<PAIR, FAST, HIGH>(a, ga, t0, st, lp, tile, tc, lane, lg);
        t0 += kPrefetch;
        if (t0 >= T) break;
        cp += gstep; if (PAIR) pp += gstep;
        ga.load(a, byte_off, t0 + kPrefetch, valid, cp, pp, cur_last, prev_last);
        pack_group<PAIR, FAST, HIGH>(a, gb, t0, st, lp, tile, tc, lane, lg);
        t0 += kPrefetch;
        if (t0 >= T) break;
    }

    if (!PAIR) {
        if (FAST) *reinterpret_cast<uint4 *>(a.state + byte_off) = st;
        else store16_bytes(a.state + byte_off, st, valid);
    }
}

template <bool PAIR, bool ALIGNED, bool HIGH>
__global__ __launch_bounds__(256) void k_diff_pack(const PackArgs a) {
    const int lane = threadIdx.x & 63;
    // one tile per wave when the grid covers the frame (the default); a smaller grid walks the tiles with its stride
    // (pipelined batches leave wave slots to the expansion of the batch before, core.hip)
    for (uint32_t tile = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); tile < a.ntiles; tile += gridDim.x * kWavesPerBlock) {
        const uint32_t tile_off = tile * kTileBytes;
        const uint32_t byte_off = tile_off + (uint32_t)lane * 16u;
        // wave-uniform choice: every lane of a full, aligned tile takes the vector path
        if (ALIGNED && tile_off + kTileBytes <= a.n) {
            pack_tile<PAIR, true, HIGH>(a, tile, byte_off, 16, lane);
        } else {
            const int valid = byte_off < a.n ? (int)min(16u, a.n - byte_off) : 0;
            pack_tile<PAIR, false, HIGH>(a, tile, byte_off, valid, lane);
        }
    }
}

hipError_t launch_diff_pack(const PackArgs &a, bool pair, bool aligned, uint32_t max_blocks, hipStream_t s) {
    const dim3 block(64 * kWavesPerBlock);
    uint32_t blocks = (a.ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    if (max_blocks && max_blocks < blocks) blocks = max_blocks;
    const dim3 grid(blocks);
    // thresholds of 128 and more (the reference's LR_THRESHOLDS is an unconstrained int, common.h:14) take the HIGH
    // form of the compare: another instantiation, the same instruction count
    const bool high = a.thr >= 128;
#define MI355_LAUNCH_PACK(P, A)                                                                        \
    do {                                                                                               \
        if (high) hipLaunchKernelGGL((k_diff_pack<P, A, true>), grid, block, 0, s, a);                 \
        else hipLaunchKernelGGL((k_diff_pack<P, A, false>), grid, block, 0, s, a);                     \
    } while (0)
    if (pair) {
        if (aligned) MI355_LAUNCH_PACK(true, true);
        else MI355_LAUNCH_PACK(true, false);
    } else {
        if (aligned) MI355_LAUNCH_PACK(false, true);
        else MI355_LAUNCH_PACK(false, false);
    }
#undef MI355_LAUNCH_PACK
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

// A *group* is kXTiles = 64 consecutive tiles (one wave of k_scan_groups reduces it); the expander's unit is a
// *range* of kWTiles = 16 tiles, and it needs, per frame, the flagged bytes in the ranges before its own.
constexpr uint32_t kXTiles = 64;          // = one wave of k_scan_groups per group
constexpr uint32_t kScanChunk = 1024;     // groups scanned per pass of k_scan_groups
constexpr uint32_t kScanDepth = 24;       // groups a wave of k_scan_groups has in flight at once (4 waves: 96 per round)

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

// grid = T, block = 256: roff[t][r] = flagged bytes of frame t in the 16-tile ranges before r (4 per group),
// totals[t] = all of them.  A wave loads the 64 byte counts of a group with one coalesced instruction and scans
// them with DPP: lane 63 gives the group's sum, lanes 15 / 31 / 47 the bytes in front of its second, third and
// fourth range; the (at most kScanChunk) group sums are scanned in LDS.
// The workgroup that finishes last (ticket counter, reset for the next launch) also scans the frame totals
// into offsets[0..T]: one launch and one dependent round trip less than a separate kernel.
__global__ __launch_bounds__(256) void k_scan_groups(const uint4 *meta, uint32_t *roff, uint32_t *totals,
                                                     uint32_t ntiles, uint32_t ngroups, uint32_t *ticket,
                                                     uint32_t *offsets) {
    static_assert(kXTiles == 64, "one wave reduces one group");
#if MI355_XPRIO
    __builtin_amdgcn_s_setprio(3);   // pipelined batches: this short kernel gates the expansion; it must not queue for
                                     // issue slots behind the next batch's pack waves
#endif
    __shared__ uint32_t s_sum[kScanChunk];
    __shared__ uint32_t s_part[kScanChunk][3];
    __shared__ uint32_t s_scan[5];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row = (size_t)blockIdx.x * ntiles;
    uint32_t *out = roff + (size_t)blockIdx.x * ngroups * 4u;
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < ngroups; g0 += kScanChunk) {
        const uint32_t gn = min(kScanChunk, ngroups - g0);
        // kScanDepth groups per wave and round, all their loads requested before the first is looked at (addresses
        // clamped, values zeroed afterwards, so that nothing waits in between): at 1080p (95 groups) ONE memory round
        // trip for the whole frame.  Beside the next batch's pack kernel (pipelined batches) a round trip takes ten
        // times as long and this kernel gates the expansion: with four groups per round it took 0.12-0.2 ms there
        for (uint32_t g = wave * kScanDepth; g < gn; g += 4u * kScanDepth) {
            uint32_t z[kScanDepth];
#pragma unroll
            for (uint32_t k = 0; k < kScanDepth; k++) {
                const uint32_t tile = (g0 + g + k) * kXTiles + lane;
                z[k] = meta[row + min(tile, ntiles - 1u)].z;
            }
#pragma unroll
            for (uint32_t k = 0; k < kScanDepth; k++) {
                const uint32_t tile = (g0 + g + k) * kXTiles + lane;
                if (g + k >= gn || tile >= ntiles) z[k] = 0u;
                const uint32_t incl = (uint32_t)wave_inclusive_scan((int)z[k]);
                if (g + k < gn) {
                    if (lane == 63) s_sum[g + k] = incl;
                    if ((lane & 15u) == 15u && lane < 48u) s_part[g + k][lane >> 4] = incl;
                }
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
            if (idx < gn)
                *reinterpret_cast<uint4 *>(out + 4 * (size_t)(g0 + idx)) =
                    make_uint4(acc, acc + s_part[idx][0], acc + s_part[idx][1], acc + s_part[idx][2]);
            acc += v[i];
        }
        carry += total;
        __syncthreads();   // s_part / s_sum are rewritten by the next chunk
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

hipError_t launch_scan(const uint4 *meta, uint32_t *roff, uint32_t *totals, uint32_t ntiles,
                       int nframes, uint32_t *offsets, uint32_t *ticket, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_groups, dim3(nframes), dim3(256), 0, s, meta, roff, totals, ntiles,
                       expand_groups(ntiles), ticket, offsets);
    return hipGetLastError();
}

// ---- expand: codes + records -> packed frame-major (xs, diff) ------------------------------------------
// grid = (ceil(W/16) rounded up to a multiple of 8, T), block = 64: one wave owns 16 consecutive tiles of
// ONE frame, i.e. one contiguous range of that frame's output.  The wave loads the meta words of its whole
// 64-tile group (the scan kernel gives the bytes in front of the group, the wave adds those in front of
// its own 16 tiles) and keeps the per-tile facts in registers (ds_bpermute).  A small LDS table maps
// "candidate r of the wave" to its tile; candidates are taken 64 at a time, four rounds to a pass: the codes
// of a pass are requested together, then the records of its multi-byte lanes, then everything is staged in
// LDS in output order (entry offsets from one DPP scan per round) and leaves with coalesced stores.
// A code with one flagged byte IS its entry; only multi-byte lanes walk the bits of their map.  No barriers.
//
// WIRE: the entries leave in the sender's byte stream instead (server/src/threads.cpp:227-229): frame t
// is {u32 n, i32 xs[n], u8 diff[n]} at byte 4t + 5*offsets[t] of a.wire, so index and payload sections
// start at arbitrary byte addresses (gfx950 global stores need no alignment).
__device__ __forceinline__ void store_u32_unaligned(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// Hand-off between lanes of ONE wave through LDS (k_expand is a single-wave workgroup): the DS operations of a
// wave execute in order, so this costs nothing in hardware; it keeps the compiler from moving LDS accesses
// across the hand-off.
__device__ __forceinline__ void lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Entries [first, first + count) of the wave, staged in LDS as (index relative to xs0) << 8 | difference, leave four
// to a lane: one ds_read_b128, one 16-byte store of the four indices and one dword of the four differences (gfx950
// global stores need no alignment: the index section is dword aligned in the packed form and byte aligned on the
// wire, the differences start at any byte).
struct __attribute__((packed, aligned(4))) U32x4A4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U32x4A1 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U32A1 { uint32_t x; };

template <bool WIRE>
__device__ __forceinline__ void flush_entries(const ExpandArgs &a, const uint32_t *stage, uint32_t first, uint32_t count,
                                              uint32_t xs0, uint32_t dst0, uint8_t *w_xs, uint8_t *w_df, size_t w_room) {
    const uint32_t lane = threadIdx.x;
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
#if MI355_XABLATE == 4
    if (xs0 != 0xfffffff0u) n = 0;
#endif
    const uint32_t n4 = n >> 2;
    for (uint32_t k = lane; k < n4; k += 64u) {
        const uint4 q = *reinterpret_cast<const uint4 *>(stage + 4 * k);
        const uint32_t x0 = xs0 + (q.x >> 8), x1 = xs0 + (q.y >> 8), x2 = xs0 + (q.z >> 8), x3 = xs0 + (q.w >> 8);
        if (WIRE) *reinterpret_cast<U32x4A1 *>(xsp + 16 * (size_t)k) = U32x4A1{x0, x1, x2, x3};
        else *reinterpret_cast<U32x4A4 *>(xsp + 16 * (size_t)k) = U32x4A4{x0, x1, x2, x3};
        // the low bytes of the four entries: v_perm picks byte 0 of two dwords at a time
        const uint32_t lo = __builtin_amdgcn_perm(q.y, q.x, 0x0c0c0400u), hi = __builtin_amdgcn_perm(q.w, q.z, 0x04000c0cu);
        *reinterpret_cast<U32A1 *>(dfp + 4 * (size_t)k) = U32A1{lo | hi};
    }
    const uint32_t tail = 4u * n4 + lane;   // the up to three entries left
    if (tail < n) {
        const uint32_t v = stage[tail];
        store_u32_unaligned(xsp + 4 * (size_t)tail, xs0 + (v >> 8));
        dfp[tail] = (uint8_t)v;
    }
}

#ifndef MI355_XROUNDS
#define MI355_XROUNDS 3
#endif
#ifndef MI355_XPRIO
#define MI355_XPRIO 2
#endif
// Ablation builds of the expander (tools/ab_build.sh, never shipped): 1 = prologue only (meta, scans), 2 = + table and
// code loads, 3 = + record loads, 4 = + staging in LDS but no output stores.  0 = the product.
#ifndef MI355_XABLATE
#define MI355_XABLATE 0
#endif
constexpr uint32_t kWTiles = 16;             // tiles per single-wave workgroup: one DPP row of lanes, a quarter of a scan group
constexpr uint32_t kWStage = 1024;     // entries staged per wave = the most a round of 64 candidates can hold
constexpr int kXRounds = MI355_XROUNDS;      // rounds of 64 candidates whose loads are requested together
#ifndef MI355_XLIGHT
#define MI355_XLIGHT 4
#endif
constexpr uint32_t kXLight = MI355_XLIGHT;   // lanes with more flagged bytes than this are expanded by 16 lanes
constexpr int kXHeavyMax = 12;               // ... unless a round of 64 candidates holds more of them than this

// What a wave needs before it can start on its item (frame t, tiles 16 sub .. 16 sub + 15).
struct ItemPro {
    uint4 m;          // lane L < 16: meta word of tile 16 sub + L
    uint32_t off_t;   // entries of the frames before t
    uint32_t roff;    // entries of frame t before the item's tiles
    uint32_t n_t;     // WIRE: entries of frame t
};

template <bool WIRE>
__device__ __forceinline__ ItemPro load_item(const ExpandArgs &a, uint32_t t, uint32_t sub, uint32_t ngroups, uint32_t lane) {
    ItemPro p;
    const uint32_t tile = sub * kWTiles + lane;
    p.m = make_uint4(0, 0, 0, 0);
    if (lane < kWTiles && tile < a.ntiles) p.m = a.meta[(size_t)t * a.ntiles + tile];
    p.off_t = a.offsets[t];
    p.roff = a.roff[(size_t)t * ngroups * 4u + sub];
    p.n_t = WIRE ? a.offsets[t + 1] - p.off_t : 0u;
    return p;
}

// inclusive scan inside the first 16 lanes (one DPP row)
__device__ __forceinline__ int row_inclusive_scan(int v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    return v;
}

// grid = (ceil(W/16), T) single-wave workgroups: one item each (the hardware's dispatcher balances the items;
// a persistent grid walking them with a fixed stride was measured 0.05 ms slower per batch: the slowest wave's
// share decides, and the waves fall into step).  8 waves per SIMD (the register budget is set for that): the
// kernel is a chain of short dependent steps, what hides them is the number of waves.
template <bool WIRE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_expand(const ExpandArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[kWTiles * 64];   // candidate rank of the wave -> tile + 1 at the first candidate of every tile, 0 elsewhere
    __shared__ uint2 s_tinfo[kWTiles];            // per tile: {byte offset of its candidate 0 in the code log - 4 * (candidates before the tile), byte offset of its first record}
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[kWStage];   // (byte index relative to the wave's first tile) << 8 | difference
#if MI355_XPRIO
    // beside the next batch's pack kernel (pipelined batches) these short, latency-bound waves must not queue for
    // issue slots behind the older, issue-hungry pack waves
    __builtin_amdgcn_s_setprio(MI355_XPRIO);
#endif
    const uint32_t lane = threadIdx.x;
    const uint32_t ngroups = (a.ntiles + kXTiles - 1) / kXTiles;
    const uint32_t t = blockIdx.y, sub = blockIdx.x;
    if (sub * kWTiles >= a.ntiles) return;   // grid.x is padded to a multiple of 8 (see launch_expand)
    const ItemPro cur = load_item<WIRE>(a, t, sub, ngroups, lane);
    {
        const uint32_t tile0 = sub * kWTiles;
        const uint4 m = cur.m;   // {code offset, record offset, flagged bytes, candidates | multi << 16}
        size_t head = 0;
        if (WIRE) {
            head = 4 * (size_t)t + 5 * (size_t)cur.off_t;
            if (sub == 0 && lane == 0 && head + 4 <= a.capacity) store_u32_unaligned(a.wire + head, cur.n_t);
        }
        // lane L < 16 <-> tile L of the item
        const uint32_t nc = m.w & 0xffffu;   // 0 in lanes >= 16
        const uint32_t rincl = (uint32_t)row_inclusive_scan((int)nc);
        const uint32_t nrec = (uint32_t)__builtin_amdgcn_readlane((int)rincl, (int)kWTiles - 1);
        if (nrec != 0) {
            const uint32_t dst0 = cur.off_t + cur.roff;   // < 2^32: the batch total is below 2^32
            const uint32_t rexcl = rincl - nc;                     // candidates of this wave before the lane's tile
            uint8_t *w_xs = nullptr, *w_df = nullptr;
            size_t w_room = 0;
            if (WIRE) {
                const size_t end = head + 4 + 5 * (size_t)cur.n_t;
                const uint32_t seg = dst0 - cur.off_t;
                w_xs = a.wire + head + 4 + 4 * (size_t)seg;
                w_df = a.wire + head + 4 + 4 * (size_t)cur.n_t + seg;
                w_room = end <= a.capacity ? (size_t)cur.n_t : 0;
            }
#if MI355_XABLATE == 1
            if (dst0 == 0xfffffff0u) a.out_xs[0] = (int32_t)nrec;
#else
            if (lane < kWTiles) s_tinfo[lane] = make_uint2(m.x - 4u * rexcl, m.y);
            // which tile candidate r of the wave belongs to: tile i owns ranks rexcl_i .. rexcl_i + nc_i - 1.  Only the
            // HEAD of every tile's range is marked (tile + 1 at rank rexcl_i, zeros elsewhere); a running maximum over
            // the candidates of a round (one DPP scan, the maximum so far carried from round to round) turns the heads
            // into "my tile" -- instead of a loop over the 16 tiles writing every rank
            reinterpret_cast<uint4 *>(s_tile)[lane] = make_uint4(0, 0, 0, 0);
            lds_handoff();
            if (lane < kWTiles && nc != 0u) s_tile[rexcl] = (uint8_t)(lane + 1u);
            lds_handoff();   // the tables are read by other lanes than the ones that wrote them
            uint32_t tile_carry = 0;   // highest head seen in the rounds before (wave-uniform)
            const uint32_t xs_base = tile0 * kTileBytes;
            uint32_t carry = 0, flushed = 0;   // entries emitted / already stored
            for (uint32_t base = 0; base < nrec; base += 64u * kXRounds) {
                // (1) the codes of the pass: table reads, then loads, back to back (indices clamped, not masked)
                uint32_t code[kXRounds], src16[kXRounds], pm[kXRounds];
                {
                    uint32_t rr[kXRounds], ti[kXRounds];
                    uint2 inf[kXRounds];
#pragma unroll
                    for (int k = 0; k < kXRounds; k++) {
                        rr[k] = min(base + 64u * (uint32_t)k + lane, nrec - 1u);
                        ti[k] = s_tile[rr[k]];
                    }
#pragma unroll
                    for (int k = 0; k < kXRounds; k++) {
                        const uint32_t run = max(wave_inclusive_max_scan(ti[k]), tile_carry);   // >= 1: candidate 0 is a head
                        tile_carry = (uint32_t)__builtin_amdgcn_readlane((int)run, 63);
                        ti[k] = run - 1u;
                    }
#pragma unroll
                    for (int k = 0; k < kXRounds; k++) inf[k] = s_tinfo[ti[k]];
#pragma unroll
                    for (int k = 0; k < kXRounds; k++) {
                        code[k] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(a.codes) + (inf[k].x + 4u * rr[k]));
                        pm[k] = inf[k].y;
                        src16[k] = ti[k] * kTileBytes;
                    }
#pragma unroll
                    for (int k = 0; k < kXRounds; k++) {
                        if (base + 64u * (uint32_t)k + lane >= nrec) code[k] = 0;   // also whole rounds beyond the wave's candidates
                        src16[k] += ((code[k] >> 24) & 63u) * 16u;                  // first byte of the lane, relative to the wave's first tile
                    }
                }
#if MI355_XABLATE == 2
                { uint32_t acc = 0;
#pragma unroll
                  for (int k = 0; k < kXRounds; k++) acc ^= code[k];
                  if (acc == 0xfffffff0u) a.out_xs[0] = (int32_t)acc; }
                continue;
#endif
                // (2) the records of its multi-byte lanes
                uint4 rec[kXRounds];
#pragma unroll
                for (int k = 0; k < kXRounds; k++) {
                    const uint32_t m16 = code[k] & 0xffffu;
                    rec[k] = make_uint4(0, 0, 0, 0);
                    if (m16 & (m16 - 1u))
                        rec[k] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.rec) + (pm[k] + ((code[k] >> 16) & 0xffu) * 16u));
                }
#if MI355_XABLATE == 3
                { uint32_t acc = 0;
#pragma unroll
                  for (int k = 0; k < kXRounds; k++) acc ^= code[k] ^ rec[k].x ^ rec[k].y ^ rec[k].z ^ rec[k].w;
                  if (acc == 0xfffffff0u) a.out_xs[0] = (int32_t)acc; }
                continue;
#endif
                // (3) stage the entries in output order
#pragma unroll
                for (int k = 0; k < kXRounds; k++) {
                    if (base + 64u * (uint32_t)k >= nrec) break;   // wave-uniform
                    const uint32_t m16 = code[k] & 0xffffu;
                    const uint32_t cnt = (uint32_t)__builtin_popcount(m16);
                    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)cnt);
                    const uint32_t round_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    if (carry - flushed + round_total > kWStage) {   // wave-uniform: make room
                        lds_handoff();
                        flush_entries<WIRE>(a, s_stage, flushed, carry - flushed, xs_base, dst0, w_xs, w_df, w_room);
                        lds_handoff();   // the stage is rewritten from its start
                        flushed = carry;
                    }
                    const uint32_t e = carry - flushed + incl - cnt;        // index in the LDS stage
                    carry += round_total;
                    // lanes with more than kXLight flagged bytes (object edges among isolated bytes) would make the whole
                    // wave walk their bits: they are expanded by 16 lanes each afterwards, four at a time
                    // -- unless the round holds more than kXHeavyMax of them (dense frames): then everybody walks, all
                    // lanes busy for as many steps as the fullest lane has bytes
                    uint64_t heavy = __ballot(cnt > kXLight);
                    const uint32_t light_max = __builtin_popcountll(heavy) > kXHeavyMax ? 16u : kXLight;
                    if (light_max == 16u) heavy = 0;
                    if (cnt == 1u) {
                        s_stage[e] = ((src16[k] + (uint32_t)__builtin_ctz(m16)) << 8) | ((code[k] >> 16) & 0xffu);   // kernels.cu:314-315
                    } else if (cnt > 1u && cnt <= light_max) {
                        uint32_t mm = m16, ee = e;
                        do {
                            const int b = __builtin_ctz(mm);
                            mm &= mm - 1;
                            const uint32_t dw = b < 8 ? (b < 4 ? rec[k].x : rec[k].y) : (b < 12 ? rec[k].z : rec[k].w);
                            s_stage[ee] = ((src16[k] + (uint32_t)b) << 8) | ((dw >> (8 * (b & 3))) & 0xffu);
                            ++ee;
                        } while (mm);
                    }
                    if (heavy) {   // wave-uniform
                        const uint32_t g = lane >> 4, b = lane & 15u;
                        uint64_t h = heavy;
                        do {
                            // the lanes of (up to) four heavy records; a missing one repeats the first and is masked out
                            const int l0 = __builtin_ctzll(h);
                            h &= h - 1;
                            const int l1 = h ? __builtin_ctzll(h) : -1;
                            h = h ? h & (h - 1) : 0;
                            const int l2 = h ? __builtin_ctzll(h) : -1;
                            h = h ? h & (h - 1) : 0;
                            const int l3 = h ? __builtin_ctzll(h) : -1;
                            h = h ? h & (h - 1) : 0;
                            const int srcl = g == 0 ? l0 : (g == 1 ? l1 : (g == 2 ? l2 : l3));
                            const int sa = (srcl < 0 ? l0 : srcl) * 4;
                            const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec[k].x);
                            const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec[k].y);
                            const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec[k].z);
                            const uint32_t r3 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec[k].w);
                            const uint32_t hm = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)m16);
                            const uint32_t he = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)e);
                            const uint32_t hs = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)src16[k]);
                            if (srcl >= 0 && ((hm >> b) & 1u)) {
                                const uint32_t pos = he + (uint32_t)__builtin_popcount(hm & ((1u << b) - 1u));
                                const uint32_t dw = b < 8 ? (b < 4 ? r0 : r1) : (b < 12 ? r2 : r3);
                                s_stage[pos] = ((hs + b) << 8) | ((dw >> (8 * (b & 3))) & 0xffu);
                            }
                        } while (h);
                    }
                }
            }
            lds_handoff();
            flush_entries<WIRE>(a, s_stage, flushed, carry - flushed, xs_base, dst0, w_xs, w_df, w_room);
#endif
        }
    }
}

hipError_t launch_expand(const ExpandArgs &a, int nframes, hipStream_t s) {
    static_assert(kWTiles * 4u == kXTiles, "k_scan_groups writes four range prefixes per group");
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
