// diff_pack.hip -- K1: per-byte |cur - state| > T, negative feedback, ordered sparse pack (gfx950).
//
// Replaces kernel2 (reference server/src/kernels.cu:289-334, launched <<<1,1024>>> at :505) and
// reproduces the CPU statement of the same step (tests/cuda_streaming/test.cu:560-576): ascending
// byte order, bounded, no atomics.
//
// Structure (HBM-bound byte streaming; no MFMA):
//   k_diff_pack : one wave64 owns one 1 KiB tile of the frame (lane l = bytes 16l..16l+15, one
//                 global_load_dwordx4 per lane per frame) for ALL frames of the batch.  In stream mode
//                 the tile's state lives in 4 VGPRs per lane across the batch, so a frame costs N
//                 bytes of HBM reads instead of 2N; frames are prefetched kPrefetch deep in a
//                 register ring.  Per frame the wave compares 16 bytes per lane, scans the per-lane
//                 counts with DPP, and appends its (index, diff) entries to the tile's private log
//                 (sequential appends: consecutive frames share cache lines, so the sparse output
//                 leaves L2 as full lines).  It also records cnt[t][tile] and logpos[t][tile].
//   k_scan_*    : per frame exclusive scan of cnt over tiles, then scan of the frame totals.
//   k_gather    : copies the log segments into the caller's packed, frame-major, ascending arrays.
// No inter-workgroup communication inside a launch, no spin waits, results independent of dispatch
// order.
#include "internal.h"

namespace mi355 {

// Ablation builds (tools/ablate.sh, never shipped): 1 = no log appends, 2 = also no cnt/logpos stores,
// 3 = loads + state fold only (memory floor of the stream loop).  0 = the product.
#ifndef MI355_ABLATE
#define MI355_ABLATE 0
#endif

constexpr int kPrefetch = 4;  // frames in flight per wave (4 x 1 KiB; ~24 waves/CU -> ~96 KiB/CU)

// ---- wave64 inclusive scan with DPP (row_shr within rows of 16, then row_bcast 15 / 31) ---------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
    return v;
}

// ---- per-dword byte arithmetic ----------------------------------------------------------------------
// 4-bit mask: bit j set iff |a.byte[j] - s.byte[j]| > thr     (kernels.cu:311-312)
__device__ __forceinline__ uint32_t dword_flags(uint32_t a, uint32_t s, int thr, uint32_t thr2) {
    uint32_t f = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int d = (int)((a >> (8 * j)) & 0xffu) - (int)((s >> (8 * j)) & 0xffu);
        f |= ((uint32_t)(d + thr) > thr2) ? (1u << j) : 0u;
    }
    return f;
}

// per-byte (a - s) mod 256                                     (kernels.cu:314 `diff[npos] = df`)
__device__ __forceinline__ uint32_t bytes_sub(uint32_t a, uint32_t s) {
    const uint32_t H = 0x80808080u;
    return ((a | H) - (s & ~H)) ^ ((a ^ ~s) & H);
}

// 4-bit mask -> 0xFF per selected byte
__device__ __forceinline__ uint32_t expand4(uint32_t f) {
    uint32_t x = (f * 0x00204081u) & 0x01010101u;
    return (x << 8) - x;
}

__device__ __forceinline__ uint4 load16_bytes(const uint8_t *p, int valid) {
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i < valid) w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void store16_bytes(uint8_t *p, uint4 v, int valid) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (i < valid) p[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
}

template <bool FAST>
__device__ __forceinline__ uint4 load16(const uint8_t *p, int valid) {
    if (FAST) return *reinterpret_cast<const uint4 *>(p);
    return load16_bytes(p, valid);
}

// One frame of one tile: compare, update the state, append to the log.  Returns the wave's count.
__device__ __forceinline__ uint32_t pack_step(const uint4 c, uint4 &s, int thr, uint32_t thr2,
                                              uint32_t byte_off, int32_t *log_xs, uint8_t *log_diff,
                                              size_t log_base) {
#if MI355_ABLATE == 3
    s.x ^= c.x; s.y ^= c.y; s.z ^= c.z; s.w ^= c.w;
    return 0;
#endif
    const uint32_t f0 = dword_flags(c.x, s.x, thr, thr2);
    const uint32_t f1 = dword_flags(c.y, s.y, thr, thr2);
    const uint32_t f2 = dword_flags(c.z, s.z, thr, thr2);
    const uint32_t f3 = dword_flags(c.w, s.w, thr, thr2);
    uint32_t m = f0 | (f1 << 4) | (f2 << 8) | (f3 << 12);

    const int cnt = __builtin_popcount(m);
    const int incl = wave_inclusive_scan(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, 63);

    if (m) {
        const uint32_t d0 = bytes_sub(c.x, s.x), d1 = bytes_sub(c.y, s.y);
        const uint32_t d2 = bytes_sub(c.z, s.z), d3 = bytes_sub(c.w, s.w);
        size_t o = log_base + (uint32_t)(incl - cnt);
#if MI355_ABLATE >= 1
        asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(o));
        m = 0;
#endif
        while (m) {
            const int j = __builtin_ctz(m);
            m &= m - 1;
            const int k = j >> 2;
            const uint32_t dw = k == 0 ? d0 : k == 1 ? d1 : k == 2 ? d2 : d3;
            log_xs[o] = (int32_t)(byte_off + (uint32_t)j);          // kernels.cu:315
            log_diff[o] = (uint8_t)(dw >> (8 * (j & 3)));           // kernels.cu:314
            ++o;
        }
        // negative feedback (kernels.cu:316-331): un-flagged bytes keep the previous value, flagged
        // bytes take the current one -> the state is the frame the client reconstructs.
        const uint32_t m0 = expand4(f0), m1 = expand4(f1), m2 = expand4(f2), m3 = expand4(f3);
        s.x = (c.x & m0) | (s.x & ~m0);
        s.y = (c.y & m1) | (s.y & ~m1);
        s.z = (c.z & m2) | (s.z & ~m2);
        s.w = (c.w & m3) | (s.w & ~m3);
    }
    return total;
}

// A group = kPrefetch consecutive frames of one tile held in registers.  Loads are always issued
// (frame index clamped to T-1) so that the number of vector-memory operations younger than any
// load is known at compile time: on gfx950 loads and stores share one in-order vmcnt, and a
// conditional load would make the compiler fall back to s_waitcnt vmcnt(0) -- i.e. no prefetch.
template <bool PAIR, bool FAST>
struct Group {
    uint4 c[kPrefetch];
    uint4 p[kPrefetch];

    __device__ __forceinline__ void load(const PackArgs &a, const uint8_t *cp, const uint8_t *pp, int t0,
                                         int valid) {
        const int last = a.nframes - 1;
#pragma unroll
        for (int d = 0; d < kPrefetch; d++) {
            const int t = min(t0 + d, last);
            c[d] = load16<FAST>(cp + (size_t)t * a.stride, valid);
            if (PAIR) p[d] = load16<FAST>(pp + (size_t)t * a.stride, valid);
        }
    }
};

template <bool PAIR, bool FAST>
__device__ __forceinline__ void pack_group(const PackArgs &a, const Group<PAIR, FAST> &g, int t0,
                                           uint4 &st, uint32_t &run, uint32_t tile, uint32_t byte_off,
                                           size_t tile_log, uint32_t thr2, int lane) {
#pragma unroll
    for (int d = 0; d < kPrefetch; d++) {
        const int t = t0 + d;
        if (t < a.nframes) {  // wave-uniform
            if (PAIR) st = g.p[d];
            const uint32_t total =
                pack_step(g.c[d], st, a.thr, thr2, byte_off, a.log_xs, a.log_diff, tile_log + run);
            if (lane == 0 && MI355_ABLATE < 2) {
                a.cnt[(size_t)t * a.ntiles + tile] = total;
                a.logpos[(size_t)t * a.ntiles + tile] = run;
            }
            run += total;
        }
    }
}

template <bool PAIR, bool FAST>
__device__ __forceinline__ void pack_tile(const PackArgs &a, uint32_t tile, uint32_t byte_off,
                                          int valid, int lane) {
    const int T = a.nframes;
    const uint8_t *cp = a.cur + byte_off;
    const uint8_t *pp = PAIR ? a.prev + byte_off : nullptr;
    const size_t tile_log = (size_t)tile * a.log_cap;
    const uint32_t thr2 = 2u * (uint32_t)a.thr;

    uint4 st = make_uint4(0, 0, 0, 0);
    if (!PAIR) st = load16<FAST>(a.state + byte_off, valid);

    // Two register groups: while one is processed (its stores are issued), the other's loads are in
    // flight; the next loads are issued right before the wait for the current ones, so the wait is an
    // exact s_waitcnt vmcnt(kPrefetch).
    Group<PAIR, FAST> ga, gb;
    uint32_t run = 0;  // entries this tile has appended to its log so far
    ga.load(a, cp, pp, 0, valid);
    for (int t0 = 0;;) {
        gb.load(a, cp, pp, t0 + kPrefetch, valid);
        pack_group<PAIR, FAST>(a, ga, t0, st, run, tile, byte_off, tile_log, thr2, lane);
        t0 += kPrefetch;
        if (t0 >= T) break;
        ga.load(a, cp, pp, t0 + kPrefetch, valid);
        pack_group<PAIR, FAST>(a, gb, t0, st, run, tile, byte_off, tile_log, thr2, lane);
        t0 += kPrefetch;
        if (t0 >= T) break;
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
// Block-wide exclusive scan of one value per thread (1024 threads = 16 waves).
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *lds /*17*/,
                                                              uint32_t &block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)v);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int w = 0; w < 16; w++) {
            const uint32_t x = lds[w];
            lds[w] = acc;
            acc += x;
        }
        lds[16] = acc;
    }
    __syncthreads();
    const uint32_t r = lds[wave] + incl - v;
    block_total = lds[16];
    __syncthreads();
    return r;
}

// grid = T, block = 1024: segoff[t][*] = exclusive scan of cnt[t][*], totals[t] = sum.
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t *cnt, uint32_t *segoff,
                                                     uint32_t *totals, uint32_t ntiles) {
    __shared__ uint32_t lds[17];
    const size_t row = (size_t)blockIdx.x * ntiles;
    const uint32_t per = (ntiles + 1023) / 1024;
    const uint32_t i0 = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; i++)
        if (i0 + i < ntiles) sum += cnt[row + i0 + i];
    uint32_t total;
    uint32_t acc = block_exclusive_scan_1024(sum, lds, total);
    for (uint32_t i = 0; i < per; i++) {
        if (i0 + i < ntiles) {
            const uint32_t c = cnt[row + i0 + i];
            segoff[row + i0 + i] = acc;
            acc += c;
        }
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// grid = 1, block = 1024: offsets[0..T] = exclusive scan of totals[0..T).
__global__ __launch_bounds__(1024) void k_scan_frames(const uint32_t *totals, uint32_t *offsets,
                                                      int nframes) {
    __shared__ uint32_t lds[17];
    uint32_t carry = 0;
    for (int base = 0; base < nframes; base += 1024) {
        const int t = base + (int)threadIdx.x;
        const uint32_t v = t < nframes ? totals[t] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, lds, total);
        if (t < nframes) offsets[t] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) offsets[nframes] = carry;
}

hipError_t launch_scan(const uint32_t *cnt, uint32_t *segoff, uint32_t *totals, uint32_t ntiles,
                       int nframes, uint32_t *offsets, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(nframes), dim3(1024), 0, s, cnt, segoff, totals, ntiles);
    hipLaunchKernelGGL(k_scan_frames, dim3(1), dim3(1024), 0, s, totals, offsets, nframes);
    return hipGetLastError();
}

// ---- gather: log segments -> packed frame-major output ----------------------------------------------
// grid = (ceil(ntiles/64), T), block = 256.  A workgroup owns 64 consecutive tiles of one frame; its
// output range is contiguous, so the writes are fully coalesced.
__global__ __launch_bounds__(256) void k_gather(const GatherArgs a) {
    __shared__ uint32_t s_incl[kGatherTiles];
    __shared__ uint32_t s_src[kGatherTiles];
    const int t = blockIdx.y;
    const uint32_t tile0 = blockIdx.x * kGatherTiles;
    const size_t row = (size_t)t * a.ntiles;

    if (threadIdx.x < kGatherTiles) {  // exactly wave 0
        const uint32_t tile = tile0 + threadIdx.x;
        const uint32_t c = tile < a.ntiles ? a.cnt[row + tile] : 0u;
        s_incl[threadIdx.x] = (uint32_t)wave_inclusive_scan((int)c);
        s_src[threadIdx.x] = tile < a.ntiles ? a.logpos[row + tile] : 0u;
    }
    __syncthreads();
    const uint32_t total = s_incl[kGatherTiles - 1];
    if (total == 0) return;
    const size_t dst0 = (size_t)a.offsets[t] + a.segoff[row + tile0];

    for (uint32_t e = threadIdx.x; e < total; e += 256) {
        // smallest sgm with s_incl[sgm] > e
        uint32_t lo = 0, hi = kGatherTiles - 1;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_incl[mid] > e) hi = mid; else lo = mid + 1;
        }
        const uint32_t sgm = lo;
        const uint32_t before = sgm ? s_incl[sgm - 1] : 0u;
        const size_t src = (size_t)(tile0 + sgm) * a.log_cap + s_src[sgm] + (e - before);
        const size_t dst = dst0 + e;
        if (dst < a.capacity) {
            a.out_xs[dst] = a.log_xs[src];
            a.out_diff[dst] = a.log_diff[src];
        }
    }
}

hipError_t launch_gather(const GatherArgs &a, int nframes, hipStream_t s) {
    const dim3 grid((a.ntiles + kGatherTiles - 1) / kGatherTiles, nframes);
    hipLaunchKernelGGL(k_gather, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace mi355
