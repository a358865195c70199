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
// The only inter-workgroup communication is the completion ticket of k_scan_groups and the tagged frame totals its last
// workgroup collects (publish_total);
// results are independent of dispatch order.
#include <cstdio>
#include <vector>

#include "pack_common.h"

namespace mi355 {

// Laboratory builds (tools/ab_build.sh: -DMI355_LAB=1 -DMI355_ABLATE=n ..., never shipped) take parts of the kernels
// out to price them; the shipped library has kAblate == kXAblate == kPad == 0 (lab.h) and none of that code.
constexpr int kStreamPrefetch = 4;   // frames per register group of the stream kernel
// Frames per register group (two groups per wave).  Stream mode: 4 (8 x 1 KiB in flight per wave, 58 VGPRs).  Pair
// mode holds two operands per frame: with 4 it needed 89 VGPRs = 5 waves per SIMD, and the 6076 waves of a 1080p frame
// no longer fitted the chip at once (5120 places): a sixth of them ran as a second generation, alone.  With 2 the
// pair kernel fits 6 waves per SIMD like the stream kernel (the depth of the prefetch was measured not to matter).
template <bool PAIR>
struct PrefetchOf { static constexpr int value = PAIR ? 2 : kStreamPrefetch; };
static_assert(kStreamPrefetch % 2 == 0, "frames are processed in pairs");

// ---- the log (round 3) ------------------------------------------------------------------------------
// What k_diff_pack leaves for k_expand, per (frame, tile):
//   * one 4-byte CODE per candidate lane (a lane with >= 1 flagged byte), in lane order:
//         bits  0..15  map of the lane's flagged bytes
//         bits 16..23  ONE flagged byte: its difference;  more: the lane's rank among the tile's
//                      multi-byte lanes of this frame (where its record is)
//         bits 24..29  the lane
//         bits 30..31  the tile modulo 4 (which tile of a round's quad or pair the expander's lane is looking at)
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
        if (kAblate == 3) { sw[k] ^= cw[k]; dm[k] = 0; fh[k] = 0; continue; }   // lab: loads + a fold of the state only
        uint32_t x;
        fh[k] = dword_flags<HIGH>(cw[k], sw[k], tc, x);
        // 0xFF in every flagged byte: a v_perm selector byte of 0x80 yields the constant 0xFF, one of 0x00
        // byte 0 of the second operand (0)
        const uint32_t mask = __builtin_amdgcn_perm(0u, 0u, fh[k]);
        dm[k] = bytes_sub_from_x(cw[k], sw[k], x, tc.h) & mask;     // diff where flagged, 0 elsewhere
        // negative feedback (kernels.cu:316-331): flagged bytes take the current value, the others
        // keep the previous one -> the state is the frame the client reconstructs
        sw[k] = bitop3<(TA & TC) | (TB & ~TC)>(cw[k], sw[k], mask);
    }
    s = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    if (kPad > 0) {   // lab: what does an instruction cost?
        uint32_t pad = cw[0];
#pragma unroll
        for (int i = 0; i < kPad; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(pad) : "v"(cw[1]));
        asm volatile("" :: "v"(pad));
    }
    // flags are 0x80 per flagged byte: two v_dot4 chains weigh them into 128 * (map of 8 bytes)
    const uint32_t lo = __builtin_amdgcn_udot4(fh[1], tc.w1, __builtin_amdgcn_udot4(fh[0], tc.w0, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(fh[3], tc.w1, __builtin_amdgcn_udot4(fh[2], tc.w0, 0u, false), false);
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
    if (nm > lp.roomM) { lp.ptrM += lp.roomM * 16u + jump; lp.roomM = 64u; }
    pm = lp.ptrM;
    if (nm == 64u) {   // wave-uniform: a DENSE tile (all 64 lanes carry two or more flagged bytes) appends records only: a
                       // lane's record is record `lane`, and its map is the record's non-zero bytes (|df| > T >= 0 is never 0)
        const u32x4 v = {dm[0], dm[1], dm[2], dm[3]};
        if (kAblate == 0) __builtin_amdgcn_raw_buffer_store_b128(v, lg.recs, pm + rankM * 16u, 0, 0);
        lp.ptrM += 64u * 16u;
        lp.roomM -= 64u;
        return 64u | (64u << 16);
    }
    if (nc > lp.roomC) { lp.ptrC += lp.roomC * 4u + jump; lp.roomC = 256u; }
    pc = lp.ptrC;
    // a lane with one flagged byte: the byte sum of its masked differences IS that byte
    const uint32_t one = __builtin_amdgcn_sad_u8((dm[0] | dm[1]) | (dm[2] | dm[3]), 0u, 0u);
    const uint32_t code = m16 | ((multi ? rankM : one) << 16) | lane24;
    if (kAblate == 0) {
        // plain (write-back) stores: the L2 is what merges a tile's 80-byte appends into whole lines (non-temporal log
        // stores were measured 2-15 % slower, profiles/README.md r04)
        __builtin_amdgcn_raw_buffer_store_b32(code, lg.codes, cand ? pc + rankC * 4u : kOOB, 0, 0);
        const u32x4 v = {dm[0], dm[1], dm[2], dm[3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, lg.recs, multi ? pm + rankM * 16u : kOOB, 0, 0);
    } else {
        asm volatile("" ::"v"(dm[0]), "v"(dm[1]), "v"(dm[2]), "v"(dm[3]), "v"(rankC), "v"(rankM), "v"(code));
    }
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

    // Aligned tiles (round 4): the group's frames through ONE buffer descriptor that starts at the group's first frame
    // and ends with the batch; voff[d] = the lane's byte offset + d * stride.  A frame costs no address arithmetic at
    // all (round 3: 8 scalar instructions to clamp the frame pointer to the last frame + a 64-bit vector add per load;
    // scalar instructions are not free on this chip: ~2.6 cycles of the SIMD's issue time each, profiles/archive/r04n), and
    // frames beyond the batch are out of the descriptor's range: they return zeros and read nothing.
    // ONCE: every frame is read once by this batch -- the frames of a stream, or pairs whose operands share no frame
    // (round-robin shards: pairs (f - 1, f) of every 8th f): non-temporal loads, +7 % for such pairs (0.355 -> 0.332 ms per
    // 128 pairs of 1080p, 4K 0.60 -> 0.63 of the roofline); pairs of CONSECUTIVE frames, where cur of one pair is prev of
    // the next, keep the plain policy (the second read hits): non-temporal loads cost them 6 % (profiles/archive/r04av).
    template <bool ONCE>
    __device__ __forceinline__ void load_desc(__amdgpu_buffer_rsrc_t cur, __amdgpu_buffer_rsrc_t prev, const uint32_t (&voff)[kPrefetch]) {
#pragma unroll
        for (int d = 0; d < kPrefetch; d++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(cur, voff[d], 0, ONCE ? 2 : 0);
            c[d] = make_uint4(v.x, v.y, v.z, v.w);
            if (PAIR) {
                const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(prev, voff[d], 0, ONCE ? 2 : 0);
                p[d] = make_uint4(w.x, w.y, w.z, w.w);
            }
        }
    }

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
    const uint32_t lane24 = ((uint32_t)lane << 24) | (tile << 30);   // bits 30..31 of a code: its tile modulo 4 (expand_group)
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
    if (kAblate < 2) {
        const uint32_t moff = (__umul24((uint32_t)t0 + (uint32_t)lane, a.ntiles) + tile) * 16u;
        const u32x4 mv = {meta.x, meta.y, meta.z, meta.w};
        __builtin_amdgcn_raw_buffer_store_b128(mv, lg.meta, (lane < kPrefetch && t0 + lane < a.nframes) ? moff : kOOB, 0, 0);
    }
}

template <bool PAIR, bool FAST, bool HIGH, bool ONCE>
__device__ __forceinline__ void pack_tile(const PackArgs &a, uint32_t tile, uint32_t byte_off,
                                          int valid, int lane) {
    const int T = a.nframes;
    // the compare constants stay in scalar registers: in vector registers the kernel was 4 % slower (profiles/archive/r04o)
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
    if (FAST) {
        uint32_t voff[kPrefetch];
#pragma unroll
        for (int d = 0; d < kPrefetch; d++) voff[d] = byte_off + (uint32_t)d * (uint32_t)a.stride;   // launch_diff_pack: (kPrefetch - 1) * stride + n < 2^32
        const uint8_t *cb = a.cur, *pb = a.prev;
        int64_t left = (int64_t)(T - 1) * (int64_t)a.stride + (int64_t)a.n;   // bytes from the group's first frame to the end of the batch's last frame
        auto desc = [&](const uint8_t *base) {
            const uint32_t bytes = left <= 0 ? 0u : (left > 0xffffffffll ? 0xffffffffu : (uint32_t)left);
            return make_rsrc(base, bytes);
        };
        ga.template load_desc<ONCE>(desc(cb), desc(pb), voff);
        for (int t0 = 0;;) {
            cb += gstep; if (PAIR) pb += gstep; left -= (int64_t)gstep;
            gb.template load_desc<ONCE>(desc(cb), desc(pb), voff);
            pack_group<PAIR, FAST, HIGH>(a, ga, t0, st, lp, tile, tc, lane, lg);
            t0 += kPrefetch;
            if (t0 >= T) break;
            cb += gstep; if (PAIR) pb += gstep; left -= (int64_t)gstep;
            ga.template load_desc<ONCE>(desc(cb), desc(pb), voff);
            pack_group<PAIR, FAST, HIGH>(a, gb, t0, st, lp, tile, tc, lane, lg);
            t0 += kPrefetch;
            if (t0 >= T) break;
        }
        if (!PAIR) *reinterpret_cast<uint4 *>(a.state + byte_off) = st;
        return;
    }
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

template <bool PAIR, bool ALIGNED, bool HIGH, bool ONCE = !PAIR>
__global__ __launch_bounds__(256) void k_diff_pack(const PackArgs a) {
    const int lane = threadIdx.x & 63;
    // one tile per wave when the grid covers the frame (the default); a smaller grid walks the tiles with its stride
    // (pipelined batches leave wave slots to the expansion of the batch before, core.hip)
    // The wave's number is the same in all its lanes, but the compiler cannot know that of threadIdx.x >> 6: without the
    // readfirstlane everything derived from the tile -- the log positions above all -- lived in VECTOR registers and was
    // updated with v_cndmask / v_add / v_mov (12 vector instructions per frame of the 114).
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (uint32_t tile = a.tile_begin + blockIdx.x * kWavesPerBlock + wave; tile < a.tile_end; tile += gridDim.x * kWavesPerBlock) {
        const uint32_t tile_off = tile * kTileBytes;
        const uint32_t byte_off = tile_off + (uint32_t)lane * 16u;
        // wave-uniform choice: every lane of a full, aligned tile takes the vector path
        if (ALIGNED && tile_off + kTileBytes <= a.n) {
            pack_tile<PAIR, true, HIGH, ONCE>(a, tile, byte_off, 16, lane);
        } else {
            const int valid = byte_off < a.n ? (int)min(16u, a.n - byte_off) : 0;
            pack_tile<PAIR, false, HIGH, ONCE>(a, tile, byte_off, valid, lane);
        }
    }
}

hipError_t launch_diff_pack(const PackArgs &a, bool pair, bool aligned, bool pair_once, uint32_t max_blocks, hipStream_t s) {
    const dim3 block(64 * kWavesPerBlock);
    uint32_t blocks = (a.tile_end - a.tile_begin + kWavesPerBlock - 1) / kWavesPerBlock;
    if (max_blocks && max_blocks < blocks) blocks = max_blocks;
    if (blocks == 0) return hipSuccess;
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
        if (aligned && pair_once) {   // operands that share no frame (core.hip, run_batch)
            if (high) hipLaunchKernelGGL((k_diff_pack<true, true, true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_diff_pack<true, true, false, true>), grid, block, 0, s, a);
        } else if (aligned) MI355_LAUNCH_PACK(true, true);
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

// How the last workgroup of k_scan_groups learns the other workgroups' frame totals (they run on different XCDs, whose
// L2s are not coherent for plain accesses) WITHOUT a release fence: a frame's total travels as ONE 64-bit atomic word
// {total, epoch of this launch}, and the reader takes a word only when its epoch is this launch's.  Value and "it has
// been written" are the same atomic object, so nothing has to be ordered between two locations: per-location coherence
// and the eventual visibility of an atomic store are all the HIP / HSA memory model is asked for.  The ticket only
// elects the reader.  (Rounds 3-4 ordered a relaxed 32-bit store before a relaxed ticket with s_waitcnt vmcnt(0), which
// gfx950's write-through agent-scope stores honour but the memory model does not promise; an agent-scope release
// fence instead writes back this XCD's whole L2 -- full of the next batch's fresh log lines when batches are pipelined
// -- 19 us instead of 12 us per launch, profiles/README.md.)  The writer has issued its store before it takes its
// ticket, so the reader's wait is bounded by that store's flight time; no workgroup waits for one that has not started.
// Round 6: the word is {total: 31 bits, tag: 33 bits} (a frame is below 2 GiB, mi355_create), and the host clears the
// totals when the tag wraps (internal.h, next_scan_epoch): the argument no longer rests on a counter's width.
__device__ __forceinline__ void publish_total(uint64_t *slot, uint32_t total, uint64_t epoch) {
    __hip_atomic_store(slot, (uint64_t)total | (epoch << kTotalBits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t read_total(const uint64_t *slot, uint64_t epoch) {
    uint64_t v;
    do v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((v >> kTotalBits) != epoch);
    return (uint32_t)v & ((1u << kTotalBits) - 1u);
}

// grid = T, block = 256: roff[t][r] = flagged bytes of frame t in the 16-tile ranges before r (4 per group),
// totals[t] = all of them.  A wave loads the 64 byte counts of a group with one coalesced instruction and scans
// them with DPP: lane 63 gives the group's sum, lanes 15 / 31 / 47 the bytes in front of its second, third and
// fourth range; the (at most kScanChunk) group sums are scanned in LDS.
// The workgroup that finishes last (ticket counter, reset for the next launch) also scans the frame totals
// into offsets[0..T]: one launch and one dependent round trip less than a separate kernel.
__global__ __launch_bounds__(256) void k_scan_groups(const uint4 *meta, uint32_t *roff, uint64_t *totals,
                                                     uint32_t ntiles, uint32_t ngroups, uint32_t *ticket,
                                                     uint64_t epoch, uint32_t *offsets, uint64_t *note) {
    static_assert(kXTiles == 64, "one wave reduces one group");
    __builtin_amdgcn_s_setprio(3);   // pipelined batches: this short kernel gates the expansion; it must not queue for
                                     // issue slots behind the next batch's pack waves
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
    // totals and the ticket are agent-scope atomics (publish_total says what orders them: nothing has to)
    __shared__ uint32_t s_is_last;
    if (threadIdx.x == 0) {
        publish_total(&totals[blockIdx.x], carry, epoch);
        s_is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_is_last) return;
    const uint32_t nframes = gridDim.x;
    carry = 0;
    for (uint32_t base = 0; base < nframes; base += 256) {
        const uint32_t t = base + threadIdx.x;
        const uint32_t v = t < nframes ? read_total(&totals[t], epoch) : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan<4>(v, s_scan, total);
        if (t < nframes) offsets[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        offsets[nframes] = carry;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the batch's total and its frames for the HOST's next scheduling decision (core.hip, adaptive overlap): one word of
        // pinned memory, one store -- a hint, ordered against nothing
        if (note) __hip_atomic_store(note, (uint64_t)carry | ((uint64_t)nframes << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

uint32_t expand_groups(uint32_t ntiles) { return (ntiles + kXTiles - 1) / kXTiles; }

hipError_t launch_scan(const uint4 *meta, uint32_t *roff, uint64_t *totals, uint32_t ntiles,
                       int nframes, uint32_t *offsets, uint32_t *ticket, uint64_t epoch, uint64_t *note, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_groups, dim3(nframes), dim3(256), 0, s, meta, roff, totals, ntiles,
                       expand_groups(ntiles), ticket, epoch, offsets, note);
    return hipGetLastError();
}

// ---- expand: codes + records -> packed frame-major (xs, diff) ------------------------------------------
// grid = (ceil(W/16), T) single-wave workgroups: a wave owns one ITEM = 16 consecutive tiles of ONE frame, i.e. one
// contiguous range of that frame's output (the scan kernel gives the entries in front of it).  Two ways through an item:
//   * the group path (expand_group<4> / <2>): items whose neighbouring four (or two) tiles have at most 64 candidates together,
//     whose entries fit the LDS stage at once and which have at most 128 lanes with two or more flagged bytes -- all
//     items of a webcam-like frame outside dense regions.  Round r expands tiles 2r and 2r + 1: the candidates of the
//     first in lanes 0 .. nc - 1, those of the second behind them, so a lane finds its tile with one compare.  The 8
//     code loads of the item are requested together straight after the meta words; the entry offsets of all rounds come
//     from eight independent DPP scans in one block of straight code; a code with one flagged byte IS its entry and is
//     staged at once; lanes with more bytes are only QUEUED (8 bytes in LDS: record index, map, stage index, source)
//     and expanded 64 at a time by all lanes afterwards -- one record load and one bit walk per item;
//   * the tile path (expand_tiles): everything else (dense regions, scene changes, synthetic worst cases) -- one
//     tile per round with the whole wave, the stage flushed whenever the next tile would not fit; a tile with all of
//     its 1024 bytes flagged bypasses the stage: its 64 records ARE the difference bytes, the indices are consecutive.
// Round 3's expander packed the candidates of all 16 tiles densely into rounds of 64 and needed a table, a
// running-maximum scan and a second table per round to find a candidate's tile again (~1150 instructions per item of
// the 1080p stream, 0.144 ms per batch; the kernel is bound by what its waves issue -- profiles/archive/r04c, r04f).
// No barriers; 5 KB of LDS and at most 64 VGPRs: 32 waves per CU, the kernel lives on its occupancy.
//
// WIRE: the entries leave in the sender's byte stream instead (server/src/threads.cpp:227-229): frame t
// is {u32 n, i32 xs[n], u8 diff[n]} at byte 4t + 5*offsets[t] of a.wire, so index and payload sections
// start at arbitrary byte addresses (gfx950 global stores need no alignment).
__device__ __forceinline__ void store_u32_unaligned(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// Hand-off between lanes of ONE wave through LDS (the waves of k_expand share nothing): the DS operations of a
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
// The packed stream is written once and not read again by this library's path: its stores are NON-TEMPORAL (round 4), so
// that they do not push the logs -- written by the pack kernel, read back here one batch later -- out of the caches.
// Measured (profiles/archive/r04an, r04ao): the batch 4-5 % faster on the slower boards of the pool (0.502 -> 0.479 ms) and
// sequentially (0.537 -> 0.514), unchanged on the fastest.  (Round 1 measured the opposite for its 4-byte + 1-byte
// scattered stores, r01d: a non-temporal store wants whole 16-byte pieces.)  The log itself is read with plain loads
// (non-temporal ones: no gain, profiles/README.md r04).
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32a1 __attribute__((aligned(1)));
template <bool BYTE_ALIGNED>
__device__ __forceinline__ void store_out4(uint8_t *p, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    const u32x4 v = {x, y, z, w};
    if (kXStore & 2) {   // lab: plain
        if (BYTE_ALIGNED) *reinterpret_cast<u32x4a1 *>(p) = v;
        else *reinterpret_cast<u32x4a4 *>(p) = v;
        return;
    }
    if (BYTE_ALIGNED) __builtin_nontemporal_store(v, reinterpret_cast<u32x4a1 *>(p));
    else __builtin_nontemporal_store(v, reinterpret_cast<u32x4a4 *>(p));
}
__device__ __forceinline__ void store_out1(uint8_t *p, uint32_t v) {   // any byte address
    if (kXStore & 1) { *reinterpret_cast<u32a1 *>(p) = v; return; }   // lab: plain
    __builtin_nontemporal_store(v, reinterpret_cast<u32a1 *>(p));
}

template <bool WIRE>
__device__ __forceinline__ void flush_entries(const ExpandArgs &a, const uint32_t *stage, uint32_t first, uint32_t count,
                                              uint32_t xs0, uint32_t dst0, uint8_t *w_xs, uint8_t *w_df, size_t w_room) {
    const uint32_t lane = threadIdx.x & 63u;
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
    const uint32_t n4 = n >> 2;
    for (uint32_t k = lane; k < n4; k += 64u) {
        const uint4 q = *reinterpret_cast<const uint4 *>(stage + 4 * k);
        const uint32_t x0 = xs0 + (q.x >> 8), x1 = xs0 + (q.y >> 8), x2 = xs0 + (q.z >> 8), x3 = xs0 + (q.w >> 8);
        store_out4<WIRE>(xsp + 16 * (size_t)k, x0, x1, x2, x3);
        // the low bytes of the four entries: v_perm picks byte 0 of two dwords at a time
        const uint32_t lo = __builtin_amdgcn_perm(q.y, q.x, 0x0c0c0400u), hi = __builtin_amdgcn_perm(q.w, q.z, 0x04000c0cu);
        store_out1(dfp + 4 * (size_t)k, lo | hi);
    }
    const uint32_t tail = 4u * n4 + lane;   // the up to three entries left
    if (tail < n) {
        const uint32_t v = stage[tail];
        store_u32_unaligned(xsp + 4 * (size_t)tail, xs0 + (v >> 8));
        dfp[tail] = (uint8_t)v;
    }
}

// The expander declares 64 vector registers although it uses 50: beside the pack kernel of the next batch (4 waves of 72
// registers per SIMD) three of its waves fit instead of four, and the pair is 1 % faster that way (0.539 against 0.546 ms
// per batch, profiles/archive/r04z: the more the expansion crowds the pack kernel, the more the pack kernel -- the longer of
// the two -- stretches).  Alone the kernel runs 8 waves per SIMD either way.
constexpr uint32_t kWTiles = 16;             // tiles per item: one DPP row of lanes, a quarter of a scan group
constexpr uint32_t kWStage = 1024;           // entries of the LDS stage = the most one tile can hold
constexpr uint32_t kFList = 128;             // most multi-byte lanes of an item on the group path (queued in 1 KiB of LDS)
constexpr uint32_t kFStage = kWStage;        // most entries of an item on the group path (10 bits of a queued lane's word)
constexpr uint32_t kXLight = 4;   // lanes with more flagged bytes than this are expanded by 16 lanes
constexpr int kXHeavyMax = 12;               // ... unless a round of 64 lanes holds more of them than this

// inclusive scan inside the first 16 lanes (one DPP row)
__device__ __forceinline__ int row_inclusive_scan(int v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    return v;
}

// The bit walk: every lane with m16 != 0 turns its record (16 masked difference bytes) into entries
// stage[e], stage[e + 1], ... = (src16 + byte) << 8 | difference.  Lanes with up to kXLight flagged bytes walk
// their bits; lanes with more (object edges among isolated bytes) would make the whole wave walk theirs, so they are
// expanded by 16 lanes each, four records at a time through ds_bpermute -- unless the wave holds more than kXHeavyMax
// of them (dense tiles): then every lane places its bytes by position.
__device__ __forceinline__ void walk_records(uint32_t m16, uint32_t e, uint32_t src16, uint4 rec, uint32_t *stage, uint32_t lane) {
    const uint32_t cnt = (uint32_t)__builtin_popcount(m16);
    const uint64_t heavy = __ballot(cnt > kXLight);
    if (__builtin_popcountll(heavy) > kXHeavyMax) {
        // DENSE wave (more than kXHeavyMax lanes with more than kXLight bytes: the synthetic worst cases, scene changes): every
        // lane places its bytes by position, straight code -- byte b of the record goes to e + (flagged bytes below b) if its
        // bit is set.  7 instructions per byte position and no loop, against 14 per iteration of the bit walk below (which
        // would run as often as the fullest lane has bytes: 14-16 times here).
        const uint32_t w[4] = {rec.x, rec.y, rec.z, rec.w};
        const uint32_t base = src16 << 8;
#pragma unroll
        for (uint32_t b = 0; b < 16u; b++) {
            const uint32_t below = (uint32_t)__builtin_popcount(m16 & ((1u << b) - 1u));
            if ((m16 >> b) & 1u) stage[e + below] = (base + (b << 8)) | ((w[b >> 2] >> (8u * (b & 3u))) & 0xffu);   // kernels.cu:314-315
        }
        return;
    }
    if (cnt != 0u && cnt <= kXLight) {
        uint32_t mm = m16, ee = e;
        do {
            const int b = __builtin_ctz(mm);
            mm &= mm - 1;
            const uint32_t dw = b < 8 ? (b < 4 ? rec.x : rec.y) : (b < 12 ? rec.z : rec.w);
            stage[ee] = ((src16 + (uint32_t)b) << 8) | ((dw >> (8 * (b & 3))) & 0xffu);   // kernels.cu:314-315
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
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec.x);
            const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec.y);
            const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec.z);
            const uint32_t r3 = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)rec.w);
            const uint32_t hm = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)m16);
            const uint32_t he = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)e);
            const uint32_t hs = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)src16);
            if (srcl >= 0 && ((hm >> b) & 1u)) {
                const uint32_t pos = he + (uint32_t)__builtin_popcount(hm & ((1u << b) - 1u));
                const uint32_t dw = b < 8 ? (b < 4 ? r0 : r1) : (b < 12 ? r2 : r3);
                stage[pos] = ((hs + b) << 8) | ((dw >> (8 * (b & 3))) & 0xffu);
            }
        } while (h);
    }
}

// The group path.  mx / my / mz hold, in lane L < 16, tile L's {byte offset of its first code, byte offset of its first
// record, entries of the item before the tile | candidates << 16}.  Round r expands the G consecutive tiles G r .. G r + G - 1
// (G = 4: "quads", G = 2: "pairs"), whose candidates together fill at most the 64 lanes: those of the first tile in lanes
// 0 .. nc0 - 1, those of the next behind them, and so on.  What a round needs of its tiles is wave-uniform and is read into
// SGPRs (v_readlane) right where it is used: through LDS the same facts cost a write, a read and a wait per round (~150
// cycles each; a third of the wave's lifetime, profiles/archive/r04l_expand_stamps.log).  Which tile of the group a candidate
// belongs to is written in its code (bits 30..31 = tile & 3, k_diff_pack): only the LOAD of the codes has to find a lane's
// tile from the counts (G - 1 compares).  A candidate's place in the output needs no tile at all: the tiles of a group
// are neighbours in the output too, so it is the entries in front of the group plus the wave-wide scan of the round.
// Round 5: quads.  On webcam-like input a tile holds ~12 candidates (isolated bytes): a pair of tiles used 24 of a
// round's 64 lanes, a quad uses 48 -- half the rounds for the same item (profiles/r05_*).
// Fills stage[0 .. entries of the item) in output order.
template <int G>
__device__ __forceinline__ void expand_group(const ExpandArgs &a, uint32_t mx, uint32_t my, uint32_t mz, uint2 *list, uint32_t *stage,
                                             uint32_t lane) {
    static_assert(G == 2 || G == 4, "a code names its tile modulo 4");
    constexpr uint32_t R = kWTiles / G;
    const __amdgpu_buffer_rsrc_t codes = make_rsrc(a.codes, a.codes_bytes), recs = make_rsrc(a.rec, a.rec_bytes);
    uint32_t code[R];
    const uint32_t lane4 = lane * 4u;
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        // lane L < t1: code L of the first tile; t1 <= L < t2: code L - t1 of the second; ... (t = running candidate count)
        uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)mz, G * r) >> 16;
        uint32_t sel = (uint32_t)__builtin_amdgcn_readlane((int)mx, G * r);
#pragma unroll
        for (uint32_t k = 1; k < G; k++) {
            const uint32_t pc = (uint32_t)__builtin_amdgcn_readlane((int)mx, G * r + k);
            sel = lane >= t ? pc - 4u * t : sel;
            t += (uint32_t)__builtin_amdgcn_readlane((int)mz, G * r + k) >> 16;
        }
        code[r] = __builtin_amdgcn_raw_buffer_load_b32(codes, lane < t ? lane4 + sel : kOOB, 0, 0);   // a lane without a candidate reads 0
    }
    if (kXAblate == 2) {   // lab: prologue + code loads
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t r = 0; r < R; r++) acc ^= code[r];
        if (acc == 0xfffffff0u) stage[0] = acc;
        return;
    }
    __builtin_amdgcn_sched_barrier(0);   // all requests leave before anything waits for the first
    // entry offsets of all rounds first: independent DPP scans in one block of straight code fill each other's
    // wait states (a scan alone is 7 dependent steps with 2 idle cycles between them)
    uint32_t ent[R];
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t cnt = (uint32_t)__builtin_popcount(code[r] & 0xffffu);
        ent[r] = (uint32_t)wave_inclusive_scan((int)cnt) - cnt;
    }
    __builtin_amdgcn_sched_barrier(0);
    // lanes with two or more flagged bytes are queued first, so that their records are on the way while the lanes
    // with one byte stage their entries
    uint32_t tail = 0;   // queued lanes (wave-uniform)
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t c = code[r];
        const uint32_t m16 = c & 0xffffu;
        const bool multi = (m16 & (m16 - 1u)) != 0u;
        const uint64_t bm = __ballot(multi);
        if (bm) {   // wave-uniform
            const uint32_t za = (uint32_t)__builtin_amdgcn_readlane((int)mz, G * r);
            const uint32_t e = (za & 0xffffu) + ent[r];
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            // {the record's rank among its tile's records | tile of the item << 28,  map | stage index << 16 | lane << 26}
            const uint32_t tile = G * r + ((c >> 30) & (G - 1u));
            if (multi) list[tail + rank] = make_uint2(((c >> 16) & 0xffu) | (tile << 28), m16 | (e << 16) | ((c & 0x3f000000u) << 2));
            tail += (uint32_t)__builtin_popcountll(bm);
        }
    }
    lds_handoff();
    // the first 64 queued lanes' records (an item rarely queues more): record index = first record of the lane's tile (lane
    // `tile` of my holds its byte offset) + rank
    const uint32_t my16 = my >> 4;
    const uint2 w0 = list[lane];
    const uint32_t base0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((w0.x >> 28) << 2), (int)my16);
    const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(recs, lane < tail ? ((base0 + (w0.x & 0xffu)) << 4) : kOOB, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t c = code[r];
        const uint32_t m16 = c & 0xffffu;
        const uint32_t za = (uint32_t)__builtin_amdgcn_readlane((int)mz, G * r);
        const uint32_t e = (za & 0xffffu) + ent[r];   // index in the stage = in the item (the group's tiles follow each other)
        // first byte of the lane relative to the item: (tile << 10) + lane of the pack kernel * 16
        const uint32_t src16 = ((c >> 20) & 0x3f0u) + ((G * r + ((c >> 30) & (G - 1u))) << 10);
        if (m16 != 0u && (m16 & (m16 - 1u)) == 0u)
            stage[e] = ((src16 + (uint32_t)__builtin_ctz(m16)) << 8) | ((c >> 16) & 0xffu);   // kernels.cu:314-315
    }
    if (kXAblate == 3) return;   // lab: + rounds
    walk_records(lane < tail ? (w0.y & 0xffffu) : 0u, (w0.y >> 16) & 0x3ffu, ((w0.x >> 28) << 10) | ((w0.y >> 26) << 4),
                 make_uint4(q0.x, q0.y, q0.z, q0.w), stage, lane);
    for (uint32_t head = 64u; head < tail; head += 64u) {   // the rest, 64 at a time
        const bool on = head + lane < tail;
        const uint2 w = list[min(head + lane, kFList - 1u)];
        const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((w.x >> 28) << 2), (int)my16);
        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(recs, on ? ((base + (w.x & 0xffu)) << 4) : kOOB, 0, 0);
        walk_records(on ? (w.y & 0xffffu) : 0u, (w.y >> 16) & 0x3ffu, ((w.x >> 28) << 10) | ((w.y >> 26) << 4),
                     make_uint4(q.x, q.y, q.z, q.w), stage, lane);
    }
}

// The tile path.  tinfo[i] = {byte offset of the first code, of the first record, flagged bytes, candidates} of the
// item's tile i; xs0 = byte index of the item's first byte, dst0 = entries of the batch before the item.
// A tile needs two dependent loads (its codes, then the records its codes name); the loop keeps the codes two tiles
// ahead and the records one tile ahead of the tile it expands, so a wave waits for memory once per item, not twice per
// tile (an item of a dense region is 16 such tiles: without the look-ahead it took ~50 us, and a scene change late in
// a batch held the whole grid up).
template <bool WIRE>
__device__ __forceinline__ void expand_tiles(const ExpandArgs &a, const uint4 *tinfo, uint32_t *stage, uint32_t lane, uint32_t xs0,
                                             uint32_t dst0, uint8_t *w_xs, uint8_t *w_df, size_t w_room) {
    const __amdgpu_buffer_rsrc_t codes = make_rsrc(a.codes, a.codes_bytes), recs = make_rsrc(a.rec, a.rec_bytes);
    // loads beyond the item (i >= 16) or without a candidate carry an offset outside the buffer: they return 0 and read nothing
    // tinfo[i].w = candidates | multi-byte lanes << 16.  A DENSE tile (64 multi-byte lanes) has records only (k_diff_pack,
    // emit_step): no codes are loaded, record `lane` is the lane's, its map are the record's non-zero bytes
    auto load_code = [&](uint32_t i) {
        const uint4 ti = tinfo[i & (kWTiles - 1u)];
        const bool dense = (ti.w >> 16) == 64u;
        return __builtin_amdgcn_raw_buffer_load_b32(codes, (i < kWTiles && !dense && lane < (ti.w & 0xffffu)) ? ti.x + 4u * lane : kOOB, 0, 0);
    };
    auto load_rec = [&](uint32_t i, uint32_t c) {
        const uint4 ti = tinfo[i & (kWTiles - 1u)];
        const uint32_t m16 = c & 0xffffu;
        const bool dense = (ti.w >> 16) == 64u;
        // a dense tile: record `lane`; otherwise the record of a lane with two or more flagged bytes
        const uint32_t off = dense ? ti.y + 16u * lane : ((m16 & (m16 - 1u)) ? ti.y + 16u * ((c >> 16) & 0xffu) : kOOB);
        return __builtin_amdgcn_raw_buffer_load_b128(recs, i < kWTiles ? off : kOOB, 0, 0);
    };
    uint32_t carry = 0, flushed = 0;   // entries of the item expanded so far / already stored (wave-uniform)
    uint32_t c0 = load_code(0), c1 = load_code(1);
    u32x4 r0 = load_rec(0, c0);
#pragma unroll 1
    for (uint32_t i = 0; i < kWTiles; i++) {
        const uint32_t c2 = load_code(i + 2u);
        const u32x4 r1 = load_rec(i + 1u, c1);
        const uint4 ti = tinfo[i];
        const uint32_t ncm = (uint32_t)__builtin_amdgcn_readfirstlane((int)ti.w), bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)ti.z);
        const uint32_t nc = ncm & 0xffffu;
        const bool dense = (ncm >> 16) == 64u;   // records only, no codes
        if (nc != 0u) {
            const bool full = bytes == kTileBytes;   // every byte of the tile flagged: its 64 records are the difference bytes
            if (full || carry - flushed + bytes > kWStage) {   // make room (a full tile goes straight out: empty the stage first)
                lds_handoff();
                flush_entries<WIRE>(a, stage, flushed, carry - flushed, xs0, dst0, w_xs, w_df, w_room);
                lds_handoff();   // the stage is rewritten from its start
                flushed = carry;
            }
            if (full && (WIRE ? w_room != 0 : (size_t)dst0 + carry + kTileBytes <= a.capacity)) {
                // all 64 lanes carry 16 bytes: record `lane` of the tile holds the differences of its bytes 16 lane .. 16 lane + 15
                // (one wave-contiguous KiB); the 1024 indices are consecutive and leave as four wave-contiguous KiB
                uint8_t *xsp = WIRE ? w_xs + 4 * (size_t)carry : (uint8_t *)(a.out_xs + (size_t)dst0 + carry);
                uint8_t *dfp = WIRE ? w_df + carry : a.out_diff + (size_t)dst0 + carry;
                const uint32_t x = xs0 + (i << 10) + 4u * lane;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    store_out4<true>(xsp + 1024 * k + 16 * lane, x + 256 * k, x + 256 * k + 1, x + 256 * k + 2, x + 256 * k + 3);
                store_out4<true>(dfp + 16 * lane, r0.x, r0.y, r0.z, r0.w);
                flushed = carry + kTileBytes;
            } else {
                const uint4 r = make_uint4(r0.x, r0.y, r0.z, r0.w);
                const uint32_t m16 = dense ? record_map16(r) : (c0 & 0xffffu);
                const uint32_t cnt = (uint32_t)__builtin_popcount(m16);
                const uint32_t e = carry - flushed + (uint32_t)wave_inclusive_scan((int)cnt) - cnt;
                const uint32_t src16 = (i << 10) + (dense ? lane * 16u : ((c0 >> 20) & 0x3f0u));
                if (cnt == 1u) stage[e] = ((src16 + (uint32_t)__builtin_ctz(m16)) << 8) | ((c0 >> 16) & 0xffu);   // kernels.cu:314-315
                const bool multi = cnt > 1u;
                if (__ballot(multi)) walk_records(multi ? m16 : 0u, e, src16, make_uint4(r0.x, r0.y, r0.z, r0.w), stage, lane);   // wave-uniform
            }
            carry += bytes;
        }
        c0 = c1; c1 = c2; r0 = r1;
    }
    lds_handoff();
    flush_entries<WIRE>(a, stage, flushed, carry - flushed, xs0, dst0, w_xs, w_df, w_room);
}

template <bool WIRE>
__global__ __launch_bounds__(64 * kXWaves) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_expand(const ExpandArgs a) {
    // kXWaves (= 1) independent waves per workgroup, an item each (nothing is shared, no barrier).  The bare dispatch of the
    // 97 280 single-wave workgroups of a 1080p batch takes 21 us, that of 24 320 four-wave workgroups 8 -- and the whole kernel
    // is 5 % SLOWER with them (117-120 -> 124-127 us alone, 130 with eight waves; pipelined batch unchanged or worse): the
    // dispatch runs beside the execution, it is not what the kernel waits for (profiles/r05i_expand_waves_per_workgroup.log)
    __shared__ __attribute__((aligned(16))) uint2 s_lists[kXWaves][kFList];         // group path: the item's queued (multi-byte) lanes; tile path: the 16 tiles' facts
    __shared__ __attribute__((aligned(16))) uint32_t s_stages[kXWaves][kWStage];    // (byte index relative to the item's first tile) << 8 | difference
    // 5120 bytes of LDS per wave: 32 waves per CU
    // beside the next batch's pack kernel (pipelined batches) these short, latency-bound waves must not queue for
    // issue slots behind the older, issue-hungry pack waves
    __builtin_amdgcn_s_setprio(2);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint2 *const s_list = s_lists[wave];
    uint32_t *const s_stage = s_stages[wave];
    if (kXAblate == 9) return;   // lab: nothing but the dispatch of the grid
    asm volatile("v_mov_b32 v63, 0" ::: "v63");   // 64 declared vector registers (above)
    const uint32_t t = blockIdx.y, sub = blockIdx.x * kXWaves + wave;
    if (sub * kWTiles >= a.ntiles) return;   // the grid is padded (see launch_expand)
    const uint32_t ngroups = (a.ntiles + kXTiles - 1) / kXTiles;
    // lane L < 16: meta word of tile 16 sub + L = {code offset, record offset, flagged bytes, candidates | multi-byte
    // lanes << 16}; the other lanes (and tiles beyond the frame) read nothing and get zeros
    const uint32_t tile = sub * kWTiles + lane;
    const __amdgpu_buffer_rsrc_t metas = make_rsrc(a.meta + (size_t)t * a.ntiles, a.ntiles * 16u);
    const u32x4 mq = __builtin_amdgcn_raw_buffer_load_b128(metas, lane < kWTiles ? tile * 16u : kOOB, 0, 0);
    // The two prefixes are wave-uniform words that only this kernel's LAST step needs (the destination of the stores): they
    // are read with SCALAR loads (constant address space: nothing writes them while this kernel runs), which leave together
    // with the meta load and are waited for on their own counter.  As plain loads the compiler placed them behind the
    // early exit below and waited for them before the code loads left: a fourth dependent round trip in a wave's life
    // (meta -> prefixes -> codes -> records).
    typedef const __attribute__((address_space(4))) uint32_t *cptr;
    const cptr offs_c = (cptr)(uintptr_t)a.offsets, roff_c = (cptr)(uintptr_t)a.roff;
    const uint32_t off_t = offs_c[t];                                           // entries of the frames before t
    const uint32_t roff = roff_c[(size_t)t * ngroups * 4u + sub];               // entries of frame t before the item's tiles
    const uint32_t n_t = WIRE ? offs_c[t + 1] - off_t : 0u;                     // entries of frame t
    asm volatile("" ::"s"(off_t), "s"(roff), "s"(n_t));   // requested HERE, beside the meta load (not behind the early exit below)
    const uint4 m = make_uint4(mq.x, mq.y, mq.z, mq.w);
    size_t head = 0;
    if (WIRE) {
        head = 4 * (size_t)t + 5 * (size_t)off_t;
        if (sub == 0 && lane == 0 && head + 4 <= a.capacity) store_u32_unaligned(a.wire + head, n_t);
    }
    const uint32_t nc = m.w & 0xffffu;   // 0 in lanes >= 16
    // the item's entries before each of its tiles | its multi-byte lanes << 16, and their totals
    const uint32_t both = m.z | (m.w & 0xffff0000u);
    const uint32_t bincl = (uint32_t)row_inclusive_scan((int)both);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)bincl, (int)kWTiles - 1);
    const uint32_t nent = tot & 0xffffu;
    if (nent == 0u) return;
    const uint32_t dst0 = off_t + roff;   // < 2^32: the batch total is below 2^32
    uint8_t *w_xs = nullptr, *w_df = nullptr;
    size_t w_room = 0;
    if (WIRE) {
        const size_t wend = head + 4 + 5 * (size_t)n_t;
        w_xs = a.wire + head + 4 + 4 * (size_t)roff;
        w_df = a.wire + head + 4 + 4 * (size_t)n_t + roff;
        w_room = wend <= a.capacity ? (size_t)n_t : 0;
    }
    const uint32_t xs0 = sub * kWTiles * kTileBytes;
    // two neighbouring tiles share a round of 64 lanes; the even tile's lane learns the odd tile's facts
    const uint32_t nc_b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nc, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
    const bool even = (lane & 1u) == 0u;
    // (a dense tile -- 64 multi-byte lanes -- has no codes: the tile path's)
    const bool pairs = __ballot((even && nc + nc_b > 64u) || (m.w >> 16) == 64u) == 0 && nent <= kFStage && (tot >> 16) <= kFList;   // wave-uniform
    if (!pairs) {
        uint4 *const s_tinfo = reinterpret_cast<uint4 *>(s_list);
        if (lane < kWTiles) s_tinfo[lane] = make_uint4(m.x, m.y, m.z, m.w);   // .w = candidates | multi-byte lanes << 16
        lds_handoff();
        expand_tiles<WIRE>(a, s_tinfo, s_stage, lane, xs0, dst0, w_xs, w_df, w_room);
        return;
    }
    if (kXAblate == 1) return;   // lab: prologue only
    // quads when every four neighbouring tiles fill at most the 64 lanes of a round (all items of a webcam-like frame away
    // from moving objects), pairs otherwise
    const uint32_t nc_q = nc_b + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(nc + nc_b), 0x102 /* row_shl:2 */, 0xf, 0xf, true);
    const bool quads = __ballot((lane & 3u) == 0u && nc + nc_q > 64u) == 0;   // wave-uniform
    if (quads) expand_group<4>(a, m.x, m.y, ((bincl - both) & 0xffffu) | (nc << 16), s_list, s_stage, lane);
    else expand_group<2>(a, m.x, m.y, ((bincl - both) & 0xffffu) | (nc << 16), s_list, s_stage, lane);
    lds_handoff();
    if (kXAblate < 2) flush_entries<WIRE>(a, s_stage, 0u, nent, xs0, dst0, w_xs, w_df, w_room);
}

hipError_t launch_expand(const ExpandArgs &a, int nframes, hipStream_t s) {
    static_assert(kWTiles * 4u == kXTiles, "k_scan_groups writes four range prefixes per group");
    // Workgroups go to the 8 XCDs round-robin by linear id: with grid.x a multiple of 8 the items of one tile range
    // land on the same XCD for every frame (the padding workgroups return at once).  It matters: a 128-byte line of a
    // tile's code log holds the codes of two consecutive frames, a line of its record log those of two or three, and the
    // L2s of the XCDs do not share -- without the padding the expander's requests to memory rise by 78 % (TCC_EA0_RDREQ
    // 1.87 M -> 3.33 M per batch, L2 hit rate 55 % -> 38 %, profiles/archive/r04_tcc_grid_padding.txt).
    const uint32_t gx = ((a.ntiles + kWTiles - 1) / kWTiles + kXWaves - 1) / kXWaves;   // workgroups per frame
    const dim3 grid((gx + 7u) / 8u * 8u, nframes);
    if (a.wire)
        hipLaunchKernelGGL(k_expand<true>, grid, dim3(64 * kXWaves), 0, s, a);
    else
        hipLaunchKernelGGL(k_expand<false>, grid, dim3(64 * kXWaves), 0, s, a);
    return hipGetLastError();
}

}  // namespace mi355
