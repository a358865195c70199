// filters.hip -- the filter kernels that sit on the same frame as the diff (gfx950).
//
// Each kernel cites the reference kernel it replaces (server/src/kernels.cu) and the CPU statement
// whose results it reproduces bit for bit.  All are byte-streaming, HBM-bound kernels: 16 pixels
// (48 B = three 16-B loads) per lane, grid-stride free (one pass), no atomics except the histogram.
// This file is compiled with -ffp-contract=off: the weighted grayscale and the convolution depend on
// separate multiply and add roundings.
#include <mutex>

#include "internal.h"

namespace mi355 {

// ---- helpers ------------------------------------------------------------------------------------
struct Px16 {  // 16 BGR pixels = 48 bytes = 12 dwords
    uint32_t w[12];
};

template <bool FAST>
__device__ __forceinline__ Px16 load_px16(const uint8_t *p, int nbytes) {
    Px16 r;
    if (FAST) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
        const uint4 a = q[0], b = q[1], c = q[2];
        r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
        r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
        r.w[8] = c.x; r.w[9] = c.y; r.w[10] = c.z; r.w[11] = c.w;
    } else {
#pragma unroll
        for (int i = 0; i < 12; i++) r.w[i] = 0;
#pragma unroll
        for (int i = 0; i < 48; i++)
            if (i < nbytes) r.w[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
    }
    return r;
}

// The colour frames of the filters are read with plain loads: non-temporal ones were measured 10 % SLOWER (gray 2.07 ->
// 2.37 us per frame, profiles/archive/r04au): a 128-byte line is touched by three of these loads, 16 bytes per lane at a 48-byte
// stride, and a non-temporal line does not wait in the cache for the other two.
__device__ __forceinline__ Px16 load_px16_once(const uint8_t *p) { return load_px16<true>(p, 48); }

template <bool FAST>
__device__ __forceinline__ void store_px16(uint8_t *p, const Px16 &r, int nbytes) {
    if (FAST) {
        uint4 *q = reinterpret_cast<uint4 *>(p);
        q[0] = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
        q[1] = make_uint4(r.w[4], r.w[5], r.w[6], r.w[7]);
        q[2] = make_uint4(r.w[8], r.w[9], r.w[10], r.w[11]);
    } else {
#pragma unroll
        for (int i = 0; i < 48; i++)
            if (i < nbytes) p[i] = (uint8_t)(r.w[i >> 2] >> (8 * (i & 3)));
    }
}

// The same 48 bytes per lane, but leaving as three WAVE-CONTIGUOUS 1 KiB stores: the lanes' 48-byte pieces are
// regrouped through the wave's 3 KiB of LDS (lane l then stores bytes 16 l + 1024 j, j = 0..2, of the wave's block).
// A store instruction whose lanes write 16 bytes at a 48-byte stride touches 24 cache lines a third each; three of
// those per wave write every line three times over.  `wave_base` = address of the wave's first byte; every lane of
// the wave must take part (caller checks with a ballot).  LDS: ds_write_b128 at a 48-byte stride is conflict-free
// per 16 lanes (12 l mod 64 covers the 64 banks in 16 disjoint groups of 4).
// ... and they are NON-TEMPORAL stores (round 4): a visualiser frame is written once and read by nobody on this path; kept
// out of the caches it leaves them to the frames and logs of the diff that follows (fused gray+binarize 2.90 -> 2.74 us per
// 1080p frame, config 3's chain 4.9 -> 4.75-4.85; the noise filter's output, which the diff reads next, gains nothing:
// profiles/archive/r04at_filters_nt_stores.log).
__device__ __forceinline__ void store_px16_wave(uint8_t *wave_base, const Px16 &r, uint4 *lds /* 192 per wave */) {
    const uint32_t lane = threadIdx.x & 63u;
    lds[lane * 3u + 0u] = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
    lds[lane * 3u + 1u] = make_uint4(r.w[4], r.w[5], r.w[6], r.w[7]);
    lds[lane * 3u + 2u] = make_uint4(r.w[8], r.w[9], r.w[10], r.w[11]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint4 *q = reinterpret_cast<uint4 *>(wave_base);
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const uint4 t = lds[j * 64u + lane];
        __builtin_nontemporal_store(v4{t.x, t.y, t.z, t.w}, reinterpret_cast<v4 *>(q + j * 64u + lane));
    }
}

// (The same regrouping on the LOAD side -- three wave-contiguous 1 KiB loads, the lanes' 48-byte pieces read back from
// LDS -- was measured and is not done: no gain for gray / binarize / red, the heat map slower (its LUT also lives in LDS):
// the caches already serve the strided loads, profiles/archive/r03t_filters_lds_ab.log.)
__device__ __forceinline__ Px16 load_px16_full(const uint8_t *in, size_t off) { return load_px16<true>(in + off, 48); }

// A lane's 48 bytes at out + off, through the wave-contiguous form when every lane of the wave stores a full piece.
// `take` = this lane takes part (full 16 pixels, 16-byte aligned frame); lanes that do not fall back by themselves.
__device__ __forceinline__ void store_px16_full(uint8_t *out, size_t off, const Px16 &q) {
    __shared__ uint4 s_t[4][192];
    if (__ballot(true) == ~0ull) {   // the whole wave is here
        store_px16_wave(out + off - (size_t)(threadIdx.x & 63u) * 48u, q, s_t[threadIdx.x >> 6]);
        return;
    }
    store_px16<true>(out + off, q, 48);
}

__device__ __forceinline__ uint32_t get_byte(const Px16 &r, int i) {  // i is a compile-time constant
    return (r.w[i >> 2] >> (8 * (i & 3))) & 0xffu;
}

__device__ __forceinline__ void put_byte(Px16 &r, int i, uint32_t v) {
    r.w[i >> 2] |= v << (8 * (i & 3));
}

// Launch shape shared by the per-pixel kernels: one lane = 16 pixels.
static inline dim3 px16_grid(uint32_t npix) {
    const uint32_t lanes = (npix + 15) / 16;
    return dim3((lanes + 255) / 256);
}

static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// ---- integer difference: tests/algorithms_benchmarks.cu:24-30 (kernel1) ------------------------------
__global__ __launch_bounds__(256) void k_int_diff(const int32_t *cur, const int32_t *prev, int32_t *out,
                                                  size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n && ((((uintptr_t)cur | (uintptr_t)prev | (uintptr_t)out) & 15u) == 0)) {
        const int4 a = *reinterpret_cast<const int4 *>(cur + i);
        const int4 b = *reinterpret_cast<const int4 *>(prev + i);
        *reinterpret_cast<int4 *>(out + i) = make_int4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    } else {
        for (size_t j = i; j < n && j < i + 4; j++) out[j] = cur[j] - prev[j];
    }
}

hipError_t launch_int_diff(const int32_t *cur, const int32_t *prev, int32_t *out, size_t n,
                           hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t lanes = (n + 3) / 4;
    hipLaunchKernelGGL(k_int_diff, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s, cur, prev,
                       out, n);
    return hipGetLastError();
}

// ---- grayscale: kernels.cu:31-43 (avg) / :67-95 (weighted) --------------------------------------------
// avg      == server/src/server.cpp:96-101            s = (B+G+R)/3
// weighted == tests/grayscale-weighted/cpu.cu:40      (uint8)(0.114*B + 0.587*G + 0.299*R) in double,
//             left to right, no contraction (the reference GPU kernel's float accumulator rounds
//             differently and is not the oracle).
// ---- weighted gray in integers ------------------------------------------------------------------------------------
// The oracle is the double expression of tests/grayscale-weighted/cpu.cu:40, uint8(0.114*B + 0.587*G + 0.299*R),
// evaluated left to right and truncated.  With K = 114 B + 587 G + 299 R the exact value is K / 1000; it is at least
// 0.001 away from an integer unless 1000 | K, far more than the double evaluation's error, so the result is
// floor(K / 1000) -- except for some of the 16 774 triples with 1000 | K, where the three rounded products add up to
// just below the integer and the truncation lands one lower: 1957 triples (tests/test_oracle.py counts them).
// For a given (B, G) at most one R in 0..255 makes K a multiple of 1000 (299 is invertible mod 1000), so the
// exceptions fit a 256 x 256 table of that R (0xFFFF: none), built on the host WITH the double expression
// (fill_gray_exceptions) -- 128 KB, resident in L2; only the one pixel in a thousand whose K is a multiple of 1000
// looks at it.  floor(K / 1000) = ((K >> 3) * 33555) >> 22 for K <= 255 000 (33555 = ceil(2^22 / 125), error
// K/8 * (33555 * 125 - 2^22) / (125 * 2^22) < 1/125).  9 integer instructions per pixel instead of 3 conversions,
// 3 multiplications and 2 additions in fp64 (every one of them in the slow instruction class) and a conversion back.
// Proven on all 2^24 triples: tests/test_oracle.py::test_integer_gray_equals_the_double_expression (CPU),
// tests/test_filters_gpu.py::test_gray_weighted_exhaustive_2_24 (device).
__device__ uint16_t g_gray_exc[65536];

// Measured (profiles/archive/r03w_gray_integer_ab.log): no faster -- the gray kernels run at the memory system's rate either
// way (2.15 us per 1080p frame), and the histogram pass of the fused chain is 4 % SLOWER in integers (3.37 against
// 3.25 us for the chain): the rare table look-up is a dependent L2 access in the middle of a streaming kernel.
// The product therefore keeps the fp64 form; -DMI355_GRAY_FP64=0 builds the integer one (bit-exact: the exhaustive
// device test passes on it).
#ifndef MI355_GRAY_FP64
#define MI355_GRAY_FP64 1
#endif
template <bool WEIGHTED>
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) {
    if (WEIGHTED) {
#if MI355_GRAY_FP64
        const double v = 0.114 * (double)b + 0.587 * (double)g + 0.299 * (double)r;
        return (uint32_t)v;
#else
        const uint32_t K = __umul24(b, 114u) + __umul24(g, 587u) + __umul24(r, 299u);
        uint32_t q = ((K >> 3) * 33555u) >> 22;
        if (K == q * 1000u && (uint32_t)g_gray_exc[(b << 8) | g] == r) q -= 1u;
        return q;
#endif
    }
    return (b + g + r) / 3u;
}

// Host side: the exception table from the double expression itself.
static void fill_gray_exceptions(uint16_t *t) {
    for (uint32_t b = 0; b < 256; b++)
        for (uint32_t g = 0; g < 256; g++) {
            t[(b << 8) | g] = 0xFFFFu;
            // the R with 114 b + 587 g + 299 R = 0 (mod 1000): R = -(114 b + 587 g) * 699 (mod 1000), 299 * 699 = 1 (mod 1000)
            const uint32_t r = (1000u - (114u * b + 587u * g) % 1000u) % 1000u * 699u % 1000u;
            if (r > 255u) continue;
            const uint32_t K = 114u * b + 587u * g + 299u * r;   // a multiple of 1000
            volatile double v = 0.114 * (double)b + 0.587 * (double)g;
            v = v + 0.299 * (double)r;
            if ((uint32_t)v != K / 1000u) t[(b << 8) | g] = (uint16_t)r;
        }
}

hipError_t init_gray_table() {   // once per device (called by mi355_create with the device current)
    static uint16_t host[65536];
    static std::once_flag once;
    std::call_once(once, [] { fill_gray_exceptions(host); });
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gray_exc), host, sizeof host);
}

// Batched launches: blockIdx.y is the frame, frame f lives at base + f*stride.
template <bool WEIGHTED, bool FAST>
__global__ __launch_bounds__(256) void k_gray(const uint8_t *in, uint8_t *out, uint32_t npix, size_t stride) {
    in += (size_t)blockIdx.y * stride;
    out += (size_t)blockIdx.y * stride;
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    if (FAST && rem >= 16) {
        const Px16 p = load_px16_once(in + off);
        Px16 q;
#pragma unroll
        for (int i = 0; i < 12; i++) q.w[i] = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t v =
                gray_of<WEIGHTED>(get_byte(p, 3 * k), get_byte(p, 3 * k + 1), get_byte(p, 3 * k + 2));
            put_byte(q, 3 * k, v); put_byte(q, 3 * k + 1, v); put_byte(q, 3 * k + 2, v);
        }
        store_px16_full(out + 0, off, q);
    } else {
        const uint32_t cntpx = rem < 16 ? rem : 16;
        for (uint32_t k = 0; k < cntpx; k++) {
            const uint8_t *p = in + off + 3 * k;
            const uint8_t v = (uint8_t)gray_of<WEIGHTED>(p[0], p[1], p[2]);
            uint8_t *q = out + off + 3 * k;
            q[0] = v; q[1] = v; q[2] = v;
        }
    }
}

hipError_t launch_gray(const uint8_t *in, uint8_t *out, uint32_t npix, bool weighted, FrameBatch fb,
                       hipStream_t s) {
    if (npix == 0 || fb.nframes <= 0) return hipSuccess;
    const bool fast = aligned16(in) && aligned16(out) && fb.stride % 16 == 0;
    dim3 g = px16_grid(npix);
    g.y = (unsigned)fb.nframes;
    const dim3 b(256);
    if (weighted) {
        if (fast) hipLaunchKernelGGL((k_gray<true, true>), g, b, 0, s, in, out, npix, fb.stride);
        else hipLaunchKernelGGL((k_gray<true, false>), g, b, 0, s, in, out, npix, fb.stride);
    } else {
        if (fast) hipLaunchKernelGGL((k_gray<false, true>), g, b, 0, s, in, out, npix, fb.stride);
        else hipLaunchKernelGGL((k_gray<false, false>), g, b, 0, s, in, out, npix, fb.stride);
    }
    return hipGetLastError();
}

// ---- binarize chain: kernels.cu:138-241, CPU semantics server/src/server.cpp:103-135 -----------------
// Histogram of one sample per pixel: 16 pixels (48 B) per lane per step, 8 private LDS copies of the
// bins per wave (flat image areas would otherwise serialise 64 lanes on one bin), merged per
// workgroup, then 256 global atomics per 16384 pixels (integer adds: order-independent).  MODE 0 reads the gray
// value replicated in a gray3 frame (every 3rd byte, server.cpp:104); MODE 1/2 read the COLOUR frame
// and compute the gray value on the fly (fused chain: no gray frame is materialised).
constexpr int kHistReplicas = 4;    // private copies of the 256 bins per wave (lane & 3): 16 KB of LDS per workgroup (round 5; 8: 4 % slower on webcam frames, 2: the same as 4)
constexpr int kHistRow = 257;       // words between two copies of the bins (not 256: see k_histogram)
constexpr int kHistBlocks = 4;      // 16-pixel x 256-lane blocks per workgroup (16384 pixels)

// gray1 != nullptr (MODE 1/2): the gray value of every pixel is also kept, one byte per pixel (frame f at
// gray1 + f * gray1_stride), so that the second pass of the fused chain reads N/3 bytes instead of the colour
// frame again and does not redo the conversion.
template <int MODE /*0: gray3 in, 1: colour in + avg, 2: colour in + weighted*/, bool FAST>
__global__ __launch_bounds__(256) void k_histogram(const uint8_t *img, uint32_t npix, int32_t *hist,
                                                   size_t stride, uint8_t *gray1, size_t gray1_stride) {
    // a copy's 256 bins are kHistRow = 257 words apart: bin g of copy r lies in LDS bank (g + r) % 64, so the copies of ONE bin
    // -- all that a flat frame touches -- are in different banks (with rows of 256 words they shared one, and flat frames
    // took the pass 5.0 us instead of 2.8 however many copies there were: profiles/r05r_histogram_replicas.log)
    __shared__ int32_t bins[4 * kHistReplicas * kHistRow];
    img += (size_t)blockIdx.y * stride;
    hist += (size_t)blockIdx.y * 256;
    if (gray1) gray1 += (size_t)blockIdx.y * gray1_stride;
    int32_t *mine = bins + ((threadIdx.x >> 6) * kHistReplicas + (threadIdx.x & (kHistReplicas - 1))) * kHistRow;
    // the workgroup's kHistBlocks x 16 pixels per lane are all requested before the first is looked at (the
    // conversions and LDS atomics of one block then run while the next blocks' bytes are on their way) -- and before
    // the 32 KB of bins are cleared (round 5): the 32 LDS writes per lane and the barrier fall into the loads' flight time
    const uint32_t px0 = (blockIdx.x * kHistBlocks * 256u + threadIdx.x) * 16u;
    const bool whole = FAST && (blockIdx.x + 1u) * kHistBlocks * 256u * 16u <= npix;   // workgroup-uniform
    if (whole) {   // (workgroup-uniform: the barrier below is reached by all or none)
        Px16 p[kHistBlocks];
#pragma unroll
        for (int it = 0; it < kHistBlocks; it++)
            p[it] = MODE != 0 ? load_px16_once(img + (size_t)(px0 + it * 4096u) * 3) : load_px16<true>(img + (size_t)(px0 + it * 4096u) * 3, 48);
        __builtin_amdgcn_sched_barrier(0);
        for (int i = threadIdx.x; i < 4 * kHistReplicas * kHistRow; i += 256) bins[i] = 0;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kHistBlocks; it++) {
            uint32_t gw[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t g = MODE == 0 ? get_byte(p[it], 3 * k)
                                             : gray_of<MODE == 2>(get_byte(p[it], 3 * k), get_byte(p[it], 3 * k + 1),
                                                                  get_byte(p[it], 3 * k + 2));
                atomicAdd(&mine[g], 1);
                gw[k >> 2] |= g << (8 * (k & 3));
            }
            if (MODE != 0 && gray1)
                *reinterpret_cast<uint4 *>(gray1 + px0 + it * 4096u) = make_uint4(gw[0], gw[1], gw[2], gw[3]);
        }
    } else {
    for (int i = threadIdx.x; i < 4 * kHistReplicas * kHistRow; i += 256) bins[i] = 0;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < kHistBlocks; it++) {
        const uint32_t lane_px = ((blockIdx.x * kHistBlocks + it) * 256u + threadIdx.x) * 16u;
        if (lane_px >= npix) break;
        const uint32_t rem = npix - lane_px;
        const size_t off = (size_t)lane_px * 3;
        if (FAST && rem >= 16) {
            const Px16 p = load_px16<true>(img + off, 48);
            uint32_t gw[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t g = MODE == 0 ? get_byte(p, 3 * k)
                                             : gray_of<MODE == 2>(get_byte(p, 3 * k), get_byte(p, 3 * k + 1),
                                                                  get_byte(p, 3 * k + 2));
                atomicAdd(&mine[g], 1);
                gw[k >> 2] |= g << (8 * (k & 3));
            }
            if (MODE != 0 && gray1) *reinterpret_cast<uint4 *>(gray1 + lane_px) = make_uint4(gw[0], gw[1], gw[2], gw[3]);
        } else {
            const uint32_t cntpx = rem < 16 ? rem : 16;
            for (uint32_t k = 0; k < cntpx; k++) {
                const uint8_t *q = img + off + 3 * k;
                const uint32_t g = MODE == 0 ? q[0] : gray_of<MODE == 2>(q[0], q[1], q[2]);
                atomicAdd(&mine[g], 1);
                if (MODE != 0 && gray1) gray1[lane_px + k] = (uint8_t)g;
            }
        }
    }
    }
    __syncthreads();
    int v = 0;
#pragma unroll
    for (int r = 0; r < 4 * kHistReplicas; r++) v += bins[r * kHistRow + threadIdx.x];
    if (v) atomicAdd(&hist[threadIdx.x], v);
}

// server.cpp:108-127, the loop AS CODED on the CPU (the oracle), one wave per frame.  The coded loop keeps the
// last two bins that were >= every bin before them: on `hist[i] >= max` it shifts index_max into
// index_sec_max and takes i; its `else if (hist[i] > sec_max && hist[i] < max)` can never fire, because every
// record sets sec_max = max (simulated on random histograms in tests/test_oracle.py::test_two_max_dead_branch
// and pinned on the reference's own binary in tests/golden/ref_server_cpu_64x48.npz).  So: bin i is a *record*
// iff hist[i] >= max(hist[0..i-1]) (the maximum of nothing is -1: bin 0 always is), index_max = the last
// record, index_sec_max = the record before it (-1 if there is none).  A lane holds 4 consecutive bins; an
// exclusive max-scan over the lanes (DPP) gives every lane the maximum before its bins.  (Run by one lane as
// written, the 256 dependent iterations took 20 us per launch.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max(int v) {
    return max(v, __builtin_amdgcn_update_dpp(-1, v, CTRL, ROW_MASK, 0xf, false));   // histogram counts are >= 0
}

__global__ __launch_bounds__(64) void k_two_max_threshold(const int32_t *histogram, int32_t *thr_out) {
    const int lane = threadIdx.x;
    const int4 h = *reinterpret_cast<const int4 *>(histogram + (size_t)blockIdx.x * 256 + 4 * lane);
    const int b[4] = {h.x, h.y, h.z, h.w};
    int v = max(max(b[0], b[1]), max(b[2], b[3]));
    v = dpp_max<0x111, 0xf>(v);  // row_shr:1  -> inclusive max-scan, as wave_inclusive_scan does sums
    v = dpp_max<0x112, 0xf>(v);
    v = dpp_max<0x114, 0xf>(v);
    v = dpp_max<0x118, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);  // row_bcast:15
    v = dpp_max<0x143, 0xc>(v);  // row_bcast:31
    int before = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1: the maximum before this lane's bins
    int top = -1, second = -1;   // this lane's last two records
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (b[k] >= before) { second = top; top = 4 * lane + k; }
        before = max(before, b[k]);
    }
    const uint64_t has = __ballot(top >= 0);          // lane 0 always has bin 0
    const int l1 = 63 - __builtin_clzll(has);
    const int index_max = __builtin_amdgcn_readlane(top, l1);
    int index_sec_max = __builtin_amdgcn_readlane(second, l1);
    const uint64_t lower = has & ((1ull << l1) - 1ull);
    if (index_sec_max < 0 && lower) index_sec_max = __builtin_amdgcn_readlane(top, 63 - __builtin_clzll(lower));
    int threshold = (index_max + index_sec_max) / 2;  // server.cpp:121
    if (threshold < 50) threshold = 50;               // :122-127
    if (threshold > 200) threshold = 200;
    if (lane == 0) thr_out[blockIdx.x] = threshold;
}

// kernels.cu:222-241 / server.cpp:129-135: byte > thr ? 255 : 0, 16 bytes per lane.
__global__ __launch_bounds__(256) void k_binarize(const uint8_t *in, uint8_t *out, uint32_t nbytes,
                                                  const int32_t *thr_p, bool fast, size_t stride) {
    in += (size_t)blockIdx.y * stride;
    out += (size_t)blockIdx.y * stride;
    const uint32_t thr = (uint32_t)thr_p[blockIdx.y];
    const uint32_t off = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (off >= nbytes) return;
    if (fast && off + 16 <= nbytes) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + off);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                o |= (((w[k] >> (8 * j)) & 0xffu) > thr ? 0xffu : 0u) << (8 * j);
            r[k] = o;
        }
        *reinterpret_cast<uint4 *>(out + off) = make_uint4(r[0], r[1], r[2], r[3]);
    } else {
        for (uint32_t i = off; i < nbytes && i < off + 16; i++) out[i] = in[i] > thr ? 255 : 0;
    }
}

// Fused second pass of config 3: colour in -> gray (avg or weighted) -> > thr -> 0/255 in the 3 channels.
template <bool WEIGHTED, bool FAST>
__global__ __launch_bounds__(256) void k_gray_binarize(const uint8_t *in, uint8_t *out, uint32_t npix,
                                                       const int32_t *thr_p, size_t stride) {
    in += (size_t)blockIdx.y * stride;
    out += (size_t)blockIdx.y * stride;
    const uint32_t thr = (uint32_t)thr_p[blockIdx.y];
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    if (FAST && rem >= 16) {
        const Px16 p = load_px16_full(in, off);
        Px16 q;
#pragma unroll
        for (int i = 0; i < 12; i++) q.w[i] = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t g =
                gray_of<WEIGHTED>(get_byte(p, 3 * k), get_byte(p, 3 * k + 1), get_byte(p, 3 * k + 2));
            const uint32_t v = g > thr ? 255u : 0u;
            put_byte(q, 3 * k, v); put_byte(q, 3 * k + 1, v); put_byte(q, 3 * k + 2, v);
        }
        store_px16_full(out, off, q);
    } else {
        const uint32_t cntpx = rem < 16 ? rem : 16;
        for (uint32_t k = 0; k < cntpx; k++) {
            const uint8_t *p = in + off + 3 * k;
            const uint8_t v = gray_of<WEIGHTED>(p[0], p[1], p[2]) > thr ? 255 : 0;
            uint8_t *q = out + off + 3 * k;
            q[0] = v; q[1] = v; q[2] = v;
        }
    }
}

// Second pass of the fused chain from the kept gray bytes: 16 pixels per lane, one 16-byte load, 48 bytes out
// (gray > thr ? 255 : 0 in the three channels, kernels.cu:222-241).
template <bool FAST>
__global__ __launch_bounds__(256) void k_binarize_gray1(const uint8_t *gray1, size_t gray1_stride, uint8_t *out,
                                                        uint32_t npix, const int32_t *thr_p, size_t stride) {
    gray1 += (size_t)blockIdx.y * gray1_stride;
    out += (size_t)blockIdx.y * stride;
    const uint32_t thr = (uint32_t)thr_p[blockIdx.y];
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    if (FAST && rem >= 16) {
        const uint4 gv = *reinterpret_cast<const uint4 *>(gray1 + lane_px);
        const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
        Px16 q;
#pragma unroll
        for (int i = 0; i < 12; i++) q.w[i] = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t v = ((gw[k >> 2] >> (8 * (k & 3))) & 0xffu) > thr ? 255u : 0u;
            put_byte(q, 3 * k, v); put_byte(q, 3 * k + 1, v); put_byte(q, 3 * k + 2, v);
        }
        store_px16_full(out, off, q);
    } else {
        const uint32_t cntpx = rem < 16 ? rem : 16;
        for (uint32_t k = 0; k < cntpx; k++) {
            const uint8_t v = gray1[lane_px + k] > thr ? 255 : 0;
            uint8_t *q = out + off + 3 * k;
            q[0] = v; q[1] = v; q[2] = v;
        }
    }
}

static hipError_t launch_hist_thr(const uint8_t *img, uint32_t npix, int mode, int32_t *hist, int32_t *thr,
                                  FrameBatch fb, hipStream_t s, uint8_t *gray1 = nullptr, size_t gray1_stride = 0) {
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)fb.nframes * 256 * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    if (npix) {
        const uint32_t px_per_block = 256 * 16 * kHistBlocks;
        const dim3 g((npix + px_per_block - 1) / px_per_block, (unsigned)fb.nframes);
        const bool fast = aligned16(img) && fb.stride % 16 == 0;
#define MI355_HIST(M)                                                                                   \
    do {                                                                                                \
        if (fast) hipLaunchKernelGGL((k_histogram<M, true>), g, dim3(256), 0, s, img, npix, hist, fb.stride, gray1, gray1_stride); \
        else hipLaunchKernelGGL((k_histogram<M, false>), g, dim3(256), 0, s, img, npix, hist, fb.stride, gray1, gray1_stride);     \
    } while (0)
        if (mode == 0) MI355_HIST(0);
        else if (mode == 1) MI355_HIST(1);
        else MI355_HIST(2);
#undef MI355_HIST
    }
    hipLaunchKernelGGL(k_two_max_threshold, dim3(fb.nframes), dim3(64), 0, s, hist, thr);
    return hipGetLastError();
}

hipError_t launch_binarize_chain(const uint8_t *gray, uint8_t *out, uint32_t nbytes, int32_t *hist,
                                 int32_t *thr, FrameBatch fb, hipStream_t s) {
    if (fb.nframes <= 0) return hipSuccess;
    hipError_t e = launch_hist_thr(gray, nbytes / 3, 0, hist, thr, fb, s);
    if (e != hipSuccess) return e;
    if (nbytes) {
        const uint32_t lanes = (nbytes + 15) / 16;
        hipLaunchKernelGGL(k_binarize, dim3((lanes + 255) / 256, (unsigned)fb.nframes), dim3(256), 0, s, gray,
                           out, nbytes, thr, aligned16(gray) && aligned16(out) && fb.stride % 16 == 0, fb.stride);
    }
    return hipGetLastError();
}

// config 3 fused: colour -> (gray, histogram) ; threshold ; gray -> binarize.  With the gray bytes kept
// (gray1: npix bytes per frame, a multiple of 16 apart): N read + N/3 written, then N/3 read + N written per
// frame -- 2.67 N; without (gray1 == nullptr): the colour frame is read and converted twice, 3 N.  The unfused
// chain moves (N+N) + N/3 + (N+N).
hipError_t launch_gray_binarize_fused(const uint8_t *color, uint8_t *out, uint32_t npix, bool weighted,
                                      int32_t *hist, int32_t *thr, FrameBatch fb, hipStream_t s, uint8_t *gray1,
                                      size_t gray1_stride) {
    if (fb.nframes <= 0) return hipSuccess;
    hipError_t e = launch_hist_thr(color, npix, weighted ? 2 : 1, hist, thr, fb, s, gray1, gray1_stride);
    if (e != hipSuccess || npix == 0) return e;
    const bool fast = aligned16(color) && aligned16(out) && fb.stride % 16 == 0;
    dim3 g = px16_grid(npix);
    g.y = (unsigned)fb.nframes;
    if (gray1) {
        if (aligned16(out) && fb.stride % 16 == 0)
            hipLaunchKernelGGL((k_binarize_gray1<true>), g, dim3(256), 0, s, gray1, gray1_stride, out, npix, thr, fb.stride);
        else
            hipLaunchKernelGGL((k_binarize_gray1<false>), g, dim3(256), 0, s, gray1, gray1_stride, out, npix, thr, fb.stride);
        return hipGetLastError();
    }
    if (weighted) {
        if (fast) hipLaunchKernelGGL((k_gray_binarize<true, true>), g, dim3(256), 0, s, color, out, npix, thr, fb.stride);
        else hipLaunchKernelGGL((k_gray_binarize<true, false>), g, dim3(256), 0, s, color, out, npix, thr, fb.stride);
    } else {
        if (fast) hipLaunchKernelGGL((k_gray_binarize<false, true>), g, dim3(256), 0, s, color, out, npix, thr, fb.stride);
        else hipLaunchKernelGGL((k_gray_binarize<false, false>), g, dim3(256), 0, s, color, out, npix, thr, fb.stride);
    }
    return hipGetLastError();
}

// ---- heat map: kernels.cu:243-270, CPU tests/heat_map_benchmark/cpu.cu:19-27,54-66 -----------------
// d = |dB|+|dG|+|dR| (0..765) -> 766-entry BGR look-up table built on the host with the reference's
// exact double expression; staged in LDS.
template <bool FAST>
__global__ __launch_bounds__(256) void k_heat_map(const uint8_t *cur, const uint8_t *prev, uint8_t *out,
                                                  uint32_t npix, const uint8_t *lut, size_t stride) {
    __shared__ uint8_t s_lut[768 * 3];
    cur += (size_t)blockIdx.y * stride;
    prev += (size_t)blockIdx.y * stride;
    out += (size_t)blockIdx.y * stride;
    for (int i = threadIdx.x; i < 766 * 3; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    const bool full = FAST && rem >= 16;
    const int nb = full ? 48 : (int)(rem < 16 ? rem : 16) * 3;
    const Px16 c = full ? load_px16_full(cur, off) : load_px16<false>(cur + off, nb);
    const Px16 p = full ? load_px16_full(prev, off) : load_px16<false>(prev + off, nb);
    Px16 q;
#pragma unroll
    for (int i = 0; i < 12; i++) q.w[i] = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int d = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const int x = (int)get_byte(c, 3 * k + ch) - (int)get_byte(p, 3 * k + ch);
            d += x < 0 ? -x : x;                                   // cpu.cu:60
        }
        put_byte(q, 3 * k, s_lut[d * 3]);                          // cpu.cu:62  B
        put_byte(q, 3 * k + 1, s_lut[d * 3 + 1]);                  // cpu.cu:63  G
        put_byte(q, 3 * k + 2, s_lut[d * 3 + 2]);                  // cpu.cu:64  R
    }
    if (full) store_px16_full(out, off, q);
    else store_px16<false>(out + off, q, nb);
}

hipError_t launch_heat_map(const uint8_t *cur, const uint8_t *prev, uint8_t *out, uint32_t npix,
                           const uint8_t *lut, FrameBatch fb, hipStream_t s) {
    if (npix == 0 || fb.nframes <= 0) return hipSuccess;
    const bool fast = aligned16(cur) && aligned16(prev) && aligned16(out) && fb.stride % 16 == 0;
    dim3 g = px16_grid(npix);
    g.y = (unsigned)fb.nframes;
    if (fast) hipLaunchKernelGGL((k_heat_map<true>), g, dim3(256), 0, s, cur, prev, out, npix, lut, fb.stride);
    else hipLaunchKernelGGL((k_heat_map<false>), g, dim3(256), 0, s, cur, prev, out, npix, lut, fb.stride);
    return hipGetLastError();
}

// ---- red motion map, dense: tests/heat_map_red_benchmark/cpu.cu:38-55 (test.cu:142-168) -----------
template <bool FAST>
__global__ __launch_bounds__(256) void k_red_dense(const uint8_t *cur, const uint8_t *prev, uint8_t *out,
                                                   uint32_t npix, int thr, size_t stride) {
    cur += (size_t)blockIdx.y * stride;
    prev += (size_t)blockIdx.y * stride;
    out += (size_t)blockIdx.y * stride;
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    const bool full = FAST && rem >= 16;
    const int nb = full ? 48 : (int)(rem < 16 ? rem : 16) * 3;
    const Px16 c = full ? load_px16_full(cur, off) : load_px16<false>(cur + off, nb);
    const Px16 p = full ? load_px16_full(prev, off) : load_px16<false>(prev + off, nb);
    Px16 q;
#pragma unroll
    for (int i = 0; i < 12; i++) q.w[i] = 0;
    const uint32_t thr2 = 2u * (uint32_t)thr;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        bool f = false;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const int x = (int)get_byte(c, 3 * k + ch) - (int)get_byte(p, 3 * k + ch);
            f = f || ((uint32_t)(x + thr) > thr2);                 // cpu.cu:44
        }
        put_byte(q, 3 * k + 2, f ? 255u : 0u);                     // cpu.cu:46-52
    }
    if (full) store_px16_full(out, off, q);
    else store_px16<false>(out + off, q, nb);
}

hipError_t launch_red_dense(const uint8_t *cur, const uint8_t *prev, uint8_t *out, uint32_t npix,
                            int thr, FrameBatch fb, hipStream_t s) {
    if (npix == 0 || fb.nframes <= 0) return hipSuccess;
    const bool fast = aligned16(cur) && aligned16(prev) && aligned16(out) && fb.stride % 16 == 0;
    dim3 g = px16_grid(npix);
    g.y = (unsigned)fb.nframes;
    if (fast) hipLaunchKernelGGL((k_red_dense<true>), g, dim3(256), 0, s, cur, prev, out, npix, thr, fb.stride);
    else hipLaunchKernelGGL((k_red_dense<false>), g, dim3(256), 0, s, cur, prev, out, npix, thr, fb.stride);
    return hipGetLastError();
}

// ---- red overlap: kernels.cu:273-281 (all entries; the h_pos/nMaxThreads truncation of :514 is not
// reproduced).  Several indices of one pixel write the same byte with the same value.
__global__ __launch_bounds__(256) void k_red_overlap(uint8_t *img, const int32_t *xs,
                                                     const uint32_t *d_count, uint32_t count,
                                                     uint32_t nbytes) {
    const uint32_t n = d_count ? *d_count : count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t x = (uint32_t)xs[i];
        const uint32_t at = x + (2u - x % 3u);
        if (at < nbytes) img[at] = 255;
    }
}

hipError_t launch_red_overlap(uint8_t *img, const int32_t *xs, const uint32_t *d_count, uint32_t count,
                              uint32_t nbytes, hipStream_t s) {
    const uint32_t upper = d_count ? nbytes : count;
    if (upper == 0) return hipSuccess;
    uint32_t blocks = (upper + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_red_overlap, dim3(blocks), dim3(256), 0, s, img, xs, d_count, count, nbytes);
    return hipGetLastError();
}

// Batched form over a packed stream: frame t's red map from its own entries xs[offsets[t] .. offsets[t+1])
// (grid.y = frame).  The frames are zeroed (kernels.cu:513) or left as they are (overlap form, :517) by the
// caller.
__global__ __launch_bounds__(256) void k_red_stream(uint8_t *out, size_t stride, const uint32_t *offsets,
                                                    const int32_t *xs, uint32_t nbytes) {
    const uint32_t first = offsets[blockIdx.y], n = offsets[blockIdx.y + 1] - first;
    uint8_t *img = out + (size_t)blockIdx.y * stride;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t x = (uint32_t)xs[first + i];
        const uint32_t at = x + (2u - x % 3u);                     // kernels.cu:273-281
        if (at < nbytes) img[at] = 255;
    }
}

// The cleared form (NOISE_VISUALIZER 2: memset + red_black_map_overlap, kernels.cu:513-514) in ONE pass that only
// writes: a WAVE owns a slice of kRedSlice = 3 KiB (1024 whole pixels) of one frame, takes the frame's entries that
// fall into it (the indices of a frame are ascending; k_red_bounds has looked the slice boundaries up), builds the
// slice in its own 3 KiB of LDS (zeros, 255 in the red byte of every pixel owning an entry) and stores it with three
// wave-contiguous 1 KiB stores.  N bytes written per frame and nothing else (memset + byte scatter was 2.7 us per
// 1080p frame; the round-2 form -- 12 KiB slices per workgroup, every slice scanning the entries of eight, three
// workgroup barriers per slice -- 2.2 us).  No workgroup barrier: a wave's LDS operations execute in order.
constexpr uint32_t kRedSlice = 3072;   // 1024 pixels

// bounds[t][j] = entries of frame t below byte j * kRedSlice (j = 0 .. slices per frame): one thread per boundary,
// a plain binary search (17 dependent probes at 1080p, all boundaries of all frames at once).
__global__ __launch_bounds__(256) void k_red_bounds(const uint32_t *offsets, const int32_t *xs, uint32_t nbytes,
                                                    uint32_t nbounds, uint32_t *bounds) {
    // (a frame has at most nbytes entries: offsets that say otherwise -- a caller's bug -- must not send the search, and the
    // slice kernel behind it, over billions of entries)
    const uint32_t first = offsets[blockIdx.y], n = min(offsets[blockIdx.y + 1] - first, nbytes);
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= nbounds) return;
    const uint64_t tgt64 = (uint64_t)j * kRedSlice;
    const uint32_t target = tgt64 < nbytes ? (uint32_t)tgt64 : nbytes;
    uint32_t lo = 0, hi = n;                       // lower bound: entries with xs < target
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((uint32_t)xs[first + mid] < target) lo = mid + 1;
        else hi = mid;
    }
    bounds[(size_t)blockIdx.y * nbounds + j] = lo;
}

__global__ __launch_bounds__(256) void k_red_stream_clear(uint8_t *out, size_t stride, const uint32_t *offsets,
                                                          const int32_t *xs, uint32_t nbytes, const uint32_t *bounds,
                                                          uint32_t nslices) {
    __shared__ uint4 s_slice[4][kRedSlice / 16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t slice = blockIdx.x * 4u + wave;
    if (slice >= nslices) return;   // wave-uniform
    const uint32_t first = offsets[blockIdx.y];
    const uint32_t *bd = bounds + (size_t)blockIdx.y * (nslices + 1u) + slice;
    const uint32_t lo = bd[0], hi = bd[1];                // the entries of this slice
    const uint32_t a = slice * kRedSlice, len = min(kRedSlice, nbytes - a);
    uint4 *sl = s_slice[wave];
    uint8_t *bytes = reinterpret_cast<uint8_t *>(sl);
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) sl[j * 64u + lane] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // slices are whole pixels: x in [a, a + len) <=> the painted byte x + (2 - x % 3) in it  (kernels.cu:273-281)
    for (uint32_t i = lo + lane; i < hi; i += 64u) {
        const uint32_t x = (uint32_t)xs[first + i] - a;
        // a is a multiple of 3: (x - a) % 3 == x % 3.  Entries outside the slice (a caller-built stream that is not
        // ascending, or the uninitialised tail of a batch that overflowed its capacity) are dropped: without the
        // guard they would paint into a neighbouring wave's slice or outside the LDS array
        if (x < len) bytes[x + (2u - x % 3u)] = 255;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint8_t *img = out + (size_t)blockIdx.y * stride + a;
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        const uint32_t o = (j * 64u + lane) * 16u;
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const uint4 t = sl[j * 64u + lane];
        if (o + 16u <= len) __builtin_nontemporal_store(v4{t.x, t.y, t.z, t.w}, reinterpret_cast<v4 *>(img + o));
        else for (uint32_t k = o; k < len; k++) img[k] = bytes[k];
    }
}

uint32_t red_bounds_per_frame(uint32_t nbytes) { return (nbytes + kRedSlice - 1) / kRedSlice + 1u; }

hipError_t launch_red_stream(uint8_t *out, const uint32_t *offsets, const int32_t *xs, uint32_t nbytes, bool clear,
                             FrameBatch fb, hipStream_t s, uint32_t *bounds_scratch) {
    if (fb.nframes <= 0 || nbytes == 0) return hipSuccess;
    if (clear && bounds_scratch && aligned16(out) && fb.stride % 16 == 0) {
        const uint32_t nb = red_bounds_per_frame(nbytes), nslices = nb - 1u;
        hipLaunchKernelGGL(k_red_bounds, dim3((nb + 255u) / 256u, (unsigned)fb.nframes), dim3(256), 0, s, offsets, xs, nbytes,
                           nb, bounds_scratch);
        hipLaunchKernelGGL(k_red_stream_clear, dim3((nslices + 3u) / 4u, (unsigned)fb.nframes), dim3(256), 0, s, out,
                           fb.stride, offsets, xs, nbytes, bounds_scratch, nslices);
        return hipGetLastError();
    }
    if (clear) {
        hipError_t e = fb.stride == nbytes
                           ? hipMemsetAsync(out, 0, (size_t)fb.nframes * nbytes, s)
                           : hipMemset2DAsync(out, fb.stride, 0, nbytes, (size_t)fb.nframes, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_red_stream, dim3(64, (unsigned)fb.nframes), dim3(256), 0, s, out, fb.stride, offsets, xs, nbytes);
    return hipGetLastError();
}

// ---- 3x3 noise filter: kernels.cu:97-136 --------------------------------------------------------------
// out[y][x][c] = (uint8) sum_{i,j} k[3i+j] * in[y+i-1][x+j-1][c], zero outside the image, float
// accumulator, taps in i-major / j-minor order, one multiply then one add per tap.
// On the interleaved byte image this is a 2-D filter with a horizontal tap spacing of 3 bytes.
//
// The store `R[...] = output` of kernels.cu:131-133 converts float to uint8_t: truncation toward zero, and what
// does not fit saturates (the conversion instruction the reference's compiler emits for it clamps; v_cvt_u32_f32
// truncates, sends negatives and NaN to 0, and the minimum does the upper clamp).  Both convolution kernels and
// the oracle use this one definition, so the result does not depend on which kernel a geometry selects.  The
// server's own kernels are non-negative and normalised: no product sum leaves [0, 255] there.
__device__ __forceinline__ uint32_t f32_to_u8(float f) {
    const uint32_t v = (uint32_t)f;
    return v < 255u ? v : 255u;
}

// Four results into one dword: v_trunc_f32 (the reference's truncation toward zero) + v_cvt_pk_u8_f32 (converts the
// now integral value exactly, saturates to 0..255 -- negatives and NaN to 0 -- and drops it into its byte): two
// instructions per byte instead of conversion, minimum and shift-or.  Same results as f32_to_u8 for every float
// (tests: test_conv3x3_sharpen_and_edge_kernels_saturate_the_same_way, the 1080p / fuzz parity against the oracle).
__device__ __forceinline__ uint32_t f32x4_to_u8x4(float a, float b, float c, float d) {
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(a), 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(b), 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(c), 2u, r);
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(d), 3u, r);
}

// Sixteen results into four dwords with ONE instruction per byte: under the round-toward-zero mode v_cvt_pk_u8_f32
// truncates by itself (and saturates to 0..255, negatives and NaN to 0) -- measured for the border cases on the MI355X
// (tools/ubench/cvt_rtz.hip, profiles/r05_cvt_rtz.txt) and for every float in tests/test_filters_gpu.py's conversion
// sweep.  The mode is switched inside ONE asm statement around the sixteen conversions only: the multiplies and adds of the
// filter are data dependences of the statement or independent of it, none can be scheduled into it.
__device__ __forceinline__ void f32x16_to_u8x16_rtz(const float (&f)[16], uint32_t (&o)[4]) {
    asm volatile(
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"   // MODE[1:0], single-precision rounding: toward zero
        "s_nop 1\n\t"
        "v_cvt_pk_u8_f32 %0, %4, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %1, %8, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %2, %12, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %3, %16, 0, 0\n\t"
        "v_cvt_pk_u8_f32 %0, %5, 1, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %9, 1, %1\n\t"
        "v_cvt_pk_u8_f32 %2, %13, 1, %2\n\t"
        "v_cvt_pk_u8_f32 %3, %17, 1, %3\n\t"
        "v_cvt_pk_u8_f32 %0, %6, 2, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %10, 2, %1\n\t"
        "v_cvt_pk_u8_f32 %2, %14, 2, %2\n\t"
        "v_cvt_pk_u8_f32 %3, %18, 2, %3\n\t"
        "v_cvt_pk_u8_f32 %0, %7, 3, %0\n\t"
        "v_cvt_pk_u8_f32 %1, %11, 3, %1\n\t"
        "v_cvt_pk_u8_f32 %2, %15, 3, %2\n\t"
        "v_cvt_pk_u8_f32 %3, %19, 3, %3\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"   // back to round-to-nearest-even (the kernels' mode)
        "s_nop 1"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
        : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]), "v"(f[8]), "v"(f[9]), "v"(f[10]),
          "v"(f[11]), "v"(f[12]), "v"(f[13]), "v"(f[14]), "v"(f[15]));
}

__device__ __forceinline__ float byte_f(uint32_t dw, int b) {  // b is a compile-time constant
    return (float)((dw >> (8 * b)) & 0xffu);                   // v_cvt_f32_ubyteN
}

// k_conv3x3_strip (rows 16-byte aligned, i.e. 3*w % 16 == 0 -- 1080p and 4K; k_conv3x3_any is the byte-wise
// form for every other geometry): a lane owns a column strip of 16 bytes x kStripRows rows and walks it
// top to bottom.  An input row is converted once and serves three output rows: as the bottom row of r-1
// (taps k6..k8, after which r-1 is complete and stored), the middle row of r (k3..k5) and the top row of
// r+1 (k0..k2) -- for every output the nine multiply-then-add steps happen in the reference's i-major,
// j-minor order.  No LDS, no barriers.  SYM: k0=k2=k6=k8 and k1=k3=k5=k7 bit for bit (the reference's
// Gaussian, server.cpp:20-36): a product k*in is the same float wherever it is used.
//
// Round 5, after the first counter pass on this kernel (profiles/r05_conv_sq.txt: a wave spent 48 % of its life in
// s_waitcnt and only 11 % waiting to issue -- the kernel was bound by the LATENCY of its row loads, not by its
// arithmetic: with two waves per SIMD a row's 170 instructions take ~0.6 us, and that was all the head start the
// next row's load had):
//   * rows are requested THREE rows ahead (a ring of three rows in flight, 15 registers);
//   * a row is ONE 16-byte load per lane: the three bytes either side come from the neighbour lanes' registers
//     (DPP wave_shr:1 / wave_shl:1 of the dword next to the seam) instead of two more dword loads per lane through
//     the cache (1.34 x the frame's bytes fetched, profiles/archive/r04_filters_pmc.json); only lanes 0 and 63 fetch the
//     dword beyond the wave's 1 KiB (one load instruction, every other lane's offset is out of range and reads nothing);
//   * loads and stores go through buffer descriptors: rows outside the image are a descriptor of zero records (the
//     hardware returns zeros: kernels.cu:111-115), lanes beyond the row end carry an out-of-range offset -- no address
//     arithmetic per lane, no zeroing instructions, no divergence;
//   * every input byte is converted ONCE: the middle tap's edge-weight product k1 * centre byte IS the product the
//     corner/edge taps of the neighbouring outputs computed (same float), so the centre bytes need no second
//     conversion in their own pairing (packed fp32 operands must be even-aligned register pairs: round 1 converted the
//     16 centre bytes twice to have them in both pairings); the adds that take such an odd-aligned pair are plain
//     v_add_f32, the others packed.
constexpr int kStripRows = 30;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t cv_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);   // raw buffer, 32-bit format
}

// One fp32 add / multiply on single registers.  Written as an instruction because the compiler otherwise re-pairs these
// operations into packed ones and moves their operands into even-aligned register pairs first (33 moves per row).
__device__ __forceinline__ float add1(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float mul1(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// (106 registers: four waves per SIMD; two or three were measured 2-6 % slower, profiles/r05_conv.txt)
template <bool SYM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_conv3x3_strip(const uint8_t *in, uint8_t *out, int rowbytes, int h, const float *k9, size_t stride) {
    constexpr int NP = 8;           // output pairs per lane and row (16 bytes)
    constexpr uint32_t kOut = 0x80000000u;   // beyond every frame (mi355_create: N < 2^31); + a row offset it does not wrap
    const uint32_t lane = threadIdx.x;
    const uint32_t frame_bytes = (uint32_t)rowbytes * (uint32_t)h;
    const __amdgpu_buffer_rsrc_t src = conv_rsrc(in + (size_t)blockIdx.z * stride, frame_bytes);
    uint8_t *dst_base = out + (size_t)blockIdx.z * stride;
    // The frame's (band of kStripRows rows, 16-byte strip) pairs are dealt to the lanes in one running number: a wave's
    // lanes are neighbouring strips of a band and, where a band ends inside the wave, go on with the first strips of the
    // next band (one wave per 64 strips of a band left 24 of the sixth wave's lanes idle at 1080p: 360 strips = 5.6 waves).
    const uint32_t strips = (uint32_t)rowbytes / 16u, nbands = ((uint32_t)h + kStripRows - 1u) / kStripRows;
    const uint32_t g = blockIdx.x * 64u + lane;
    const uint32_t band = g / strips, strip = g - band * strips;
    const bool live = band < nbands;
    const uint32_t xb = strip * 16u;
    const int y0 = (int)(band * kStripRows);   // (per lane)
    // Byte offsets of row y0 - 1 (the walk's first row) in the frame: step k adds k rows.  Rows outside the image need no
    // care: row -1 is a "negative" offset, i.e. one beyond the descriptor's records, and so is every row >= h -- the
    // hardware returns zeros for them (kernels.cu:111-115) and drops their stores.  A lane without a strip, and the halo
    // fetch of a lane that needs none, start beyond the records and stay there (kOut + 40 rows does not wrap).
    const uint32_t col0 = live ? (uint32_t)((y0 - 1) * rowbytes) + xb : kOut;
    const bool has_l = live && strip > 0u, has_r = live && strip + 1u < strips;
    // the dword beyond the wave's own kilobyte: lane 0 needs the one to its left, lane 63 the one to its right (zero
    // outside the row); the other lanes read nothing
    const uint32_t edge0 = lane == 0u ? (has_l ? col0 - 4u : kOut) : (lane == 63u && has_r ? col0 + 16u : kOut);
    const uint32_t mask_l = lane == 0u ? ~0u : 0u, mask_r = lane == 63u ? ~0u : 0u;
    // where a band ends inside the wave the neighbour lane holds the other end of the image: no halo from there
    const uint32_t keep_l = has_l ? ~0u : 0u, keep_r = has_r ? ~0u : 0u;
    float kk[9];
#pragma unroll
    for (int i = 0; i < 9; i++) kk[i] = k9[i];

    struct Row { uint32_t m[4]; uint32_t edge; };
    auto load_row = [&](int k) {   // row y0 - 1 + k of the lane's band
        const uint32_t rowoff = (uint32_t)k * (uint32_t)rowbytes;
        Row v;
        const cv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(src, col0 + rowoff, 0, 0);
        v.edge = __builtin_amdgcn_raw_buffer_load_b32(src, edge0 + rowoff, 0, 0);
        v.m[0] = t.x; v.m[1] = t.y; v.m[2] = t.z; v.m[3] = t.w;
        return v;
    };
    // One input row: B = accumulators of output row r-1, M = of row r, T = (fresh) of row r+1, as NP pairs
    // of adjacent outputs.  With f[n] = byte xb - 3 + n (n = 0..21), output pair q = outputs (2q, 2q+1) takes its left
    // and right taps from e[q] = (f[2q], f[2q+1]) and e[q+3], and its middle tap from (f[2q+3], f[2q+4]) = (e[q+1].y,
    // e[q+2].x): an odd-aligned pair.
    // (`v` is refilled with row r + 3 as soon as its bytes have been converted: three rows in flight, and no register
    // copies to rotate the ring -- a copy of a row that is still on its way would wait for it)
    auto step = [&](int k, Row &v, f32x2 (&B)[NP], f32x2 (&M)[NP], f32x2 (&T)[NP]) {   // k: row y0 - 1 + k
        // the neighbour lanes' dwords next to the seam (lanes without a neighbour get 0), lanes 0 / 63: the fetched one
        uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.m[3], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        uint32_t rr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.m[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        l = ((v.edge & mask_l) | l) & keep_l;
        rr = ((v.edge & mask_r) | rr) & keep_r;
        f32x2 e[NP + 3];
        e[0] = f32x2{byte_f(l, 1), byte_f(l, 2)};
        e[1] = f32x2{byte_f(l, 3), byte_f(v.m[0], 0)};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            e[2 + 2 * d] = f32x2{byte_f(v.m[d], 1), byte_f(v.m[d], 2)};
            e[3 + 2 * d] = f32x2{byte_f(v.m[d], 3), d + 1 < 4 ? byte_f(v.m[d + 1 < 4 ? d + 1 : d], 0) : byte_f(rr, 0)};
        }
        e[NP + 2] = f32x2{byte_f(rr, 1), byte_f(rr, 2)};
        __builtin_amdgcn_sched_barrier(0);
        v = load_row(k + 3);
        __builtin_amdgcn_sched_barrier(0);
        if (SYM) {
            f32x2 pc[NP + 3], pe[NP + 3];   // corner * e, edge * e
            float kc;                       // the centre weight in a vector register (add1 / mul1 take vector operands)
            asm("v_mov_b32 %0, %1" : "=v"(kc) : "s"(kk[4]));
#pragma unroll
            for (int q = 0; q < 3; q++) { pc[q] = kk[0] * e[q]; pe[q] = kk[1] * e[q]; }
#pragma unroll
            for (int q = 0; q < NP; q++) {
                pc[q + 3] = kk[0] * e[q + 3];
                pe[q + 3] = kk[1] * e[q + 3];
                // middle taps: edge weight * centre bytes = what pe already holds; centre weight: two plain multiplies
                const float pox = pe[q + 1].y, poy = pe[q + 2].x;
                const float pmx = mul1(kc, e[q + 1].y), pmy = mul1(kc, e[q + 2].x);
                f32x2 b = B[q] + pc[q];                                // k6
                b.x = add1(b.x, pox); b.y = add1(b.y, poy);            // k7
                B[q] = b + pc[q + 3];                                  // k8
                f32x2 m = M[q] + pe[q];                                // k3
                m.x = add1(m.x, pmx); m.y = add1(m.y, pmy);            // k4
                M[q] = m + pe[q + 3];                                  // k5
                f32x2 t;
                t.x = add1(pc[q].x, pox); t.y = add1(pc[q].y, poy);    // k0 k1 (0 + p == p up to the sign of zero)
                T[q] = t + pc[q + 3];                                  // k2
            }
        } else {
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const f32x2 c = f32x2{e[q + 1].y, e[q + 2].x};
                B[q] = ((B[q] + kk[6] * e[q]) + kk[7] * c) + kk[8] * e[q + 3];
                M[q] = ((M[q] + kk[3] * e[q]) + kk[4] * c) + kk[5] * e[q + 3];
                T[q] = (kk[0] * e[q] + kk[1] * c) + kk[2] * e[q + 3];
            }
        }
        {   // output row r-1 is complete; rows outside the strip (the walk's first step, the steps that round it up to
            // whole groups of three) go to a descriptor without records and are dropped -- no branch: behind a branch the
            // compiler loses count of the loads in flight and waits for all of them at the head of the loop
            const float bf[16] = {B[0].x, B[0].y, B[1].x, B[1].y, B[2].x, B[2].y, B[3].x, B[3].y,
                                  B[4].x, B[4].y, B[5].x, B[5].y, B[6].x, B[6].y, B[7].x, B[7].y};
            uint32_t ow[4];
            f32x16_to_u8x16_rtz(bf, ow);                           // :131-133
            const cv_u32x4 o = {ow[0], ow[1], ow[2], ow[3]};
            // output row y0 - 2 + k: the band's rows are k = 2 .. kStripRows + 1 (a descriptor without records drops the
            // others); rows >= h of the image's last band are beyond the records by themselves
            const bool mine = k >= 2 && k <= kStripRows + 1;
            const __amdgpu_buffer_rsrc_t d = conv_rsrc(dst_base, mine ? frame_bytes : 0u);
            __builtin_amdgcn_raw_buffer_store_b128(o, d, col0 + (uint32_t)(k - 1) * (uint32_t)rowbytes, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // one row at a time: fewer products live at once
    };

    f32x2 a0[NP], a1[NP], a2[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) a0[q] = a1[q] = a2[q] = f32x2{0.0f, 0.0f};
    // rows y0-1 .. y1 are walked; three rows are in flight ahead of the one being computed (requests for rows beyond y1
    // read rows of the strip below or, beyond the image, nothing)
    // (requested in the order they are used: the wait at the head of the loop is the worse of what the code in front of
    // the loop and the loop's own end leave in flight, and the compiler's scheduler, left alone, asked for row y0 - 1 last)
    Row q0 = load_row(0);
    __builtin_amdgcn_sched_barrier(0);
    Row q1 = load_row(1);
    __builtin_amdgcn_sched_barrier(0);
    Row q2 = load_row(2);
    __builtin_amdgcn_sched_barrier(0);
    // rows y0 - 1 .. y0 + kStripRows in whole groups of three (the accumulator sets and the row buffers rotate through
    // their roles); the step too many computes a row nobody stores
    for (int k = 0; k <= kStripRows + 1; k += 3) {
        step(k, q0, a0, a1, a2);
        step(k + 1, q1, a1, a2, a0);
        step(k + 2, q2, a2, a0, a1);
    }
}

constexpr int kConvTW = 64;    // output pixels per band (byte-wise kernel)
constexpr int kConvRows = 8;   // output rows per workgroup (byte-wise kernel)

__global__ __launch_bounds__(256) void k_conv3x3_any(const uint8_t *in, uint8_t *out, int w, int h,
                                                     const float *k9, size_t stride) {
    __shared__ uint8_t tile[kConvRows + 2][(kConvTW + 2) * 3 + 2];
    in += (size_t)blockIdx.z * stride;
    out += (size_t)blockIdx.z * stride;
    __shared__ float sk[9];
    if (threadIdx.x < 9) sk[threadIdx.x] = k9[threadIdx.x];
    const int x0 = blockIdx.x * kConvTW, y0 = blockIdx.y * kConvRows;
    const int rowbytes = (kConvTW + 2) * 3;
    for (int i = threadIdx.x; i < (kConvRows + 2) * rowbytes; i += 256) {
        const int ry = i / rowbytes, rb = i - ry * rowbytes;
        const int gy = y0 + ry - 1;
        const int gxb = (x0 - 1) * 3 + rb;          // byte column in the image row
        uint8_t v = 0;                              // zero halo, kernels.cu:111-115
        if (gy >= 0 && gy < h && gxb >= 0 && gxb < w * 3) v = in[((size_t)gy * w) * 3 + gxb];
        tile[ry][rb] = v;
    }
    __syncthreads();
    // 64 px * 3 ch = 192 byte columns per row, 8 rows -> 1536 outputs, 6 per thread
    for (int o = threadIdx.x; o < kConvRows * kConvTW * 3; o += 256) {
        const int ry = o / (kConvTW * 3), cb = o - ry * (kConvTW * 3);
        const int gx = x0 + cb / 3, gy = y0 + ry;
        if (gx >= w || gy >= h) continue;
        float acc = 0.0f;                                          // kernels.cu:120-122
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float prod = sk[i * 3 + j] * (float)tile[ry + i][cb + j * 3];
                acc = acc + prod;                                  // kernels.cu:126-128
            }
        out[((size_t)gy * w) * 3 + (size_t)x0 * 3 + cb] = (uint8_t)f32_to_u8(acc);   // kernels.cu:131-133
    }
}

// ---- K x K filter of the reference's filter study: tests/noise_filter_benchmark/v2.cu:36-80 ------------------------
// The same operator as convolution_kernel above with K = 1..9 (even K included): output (y, x, c) sums
// k[i*K+j] * in[y - K/2 + i][x - K/2 + j][c] over i (outer) and j (inner), zero outside the image, fp32, one
// multiply then one add per tap, result truncated and saturated (f32_to_u8).  This is the study's path (it fixes
// the report's table of residual changes, REPORT/report.tex:2601-2611), not the server's: a lane owns 4 consecutive
// bytes of a row and reads its taps through the cache (one unaligned dword per tap inside the image, bytes at the
// borders); no tiling.  K = 3 on the server path is k_conv3x3_strip.
constexpr int kConvKMax = 9;

__global__ __launch_bounds__(256) void k_conv_kxk(const uint8_t *in, uint8_t *out, int rowbytes, int h, const float *kk,
                                                  int K, size_t stride) {
    __shared__ float sk[kConvKMax * kConvKMax];
    in += (size_t)blockIdx.z * stride;
    out += (size_t)blockIdx.z * stride;
    if ((int)threadIdx.x < K * K) sk[threadIdx.x] = kk[threadIdx.x];
    __syncthreads();
    const int xb = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xb >= rowbytes || y >= h) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int half = K / 2;
    for (int i = 0; i < K; i++) {
        const int yy = y - half + i;
        const bool row_in = yy >= 0 && yy < h;
        const uint8_t *row = in + (size_t)(row_in ? yy : 0) * rowbytes;
        for (int j = 0; j < K; j++) {
            const float kv = sk[i * K + j];
            const int col = xb + 3 * (j - half);                     // byte column of the tap of output byte xb
            uint32_t dw = 0;
            if (row_in) {
                if (col >= 0 && col + 4 <= rowbytes) __builtin_memcpy(&dw, row + col, 4);
                else
                    for (int b = 0; b < 4; b++)
                        if (col + b >= 0 && col + b < rowbytes) dw |= (uint32_t)row[col + b] << (8 * b);   // zero halo, v2.cu:46-54
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const float prod = kv * (float)((dw >> (8 * b)) & 0xffu);
                acc[b] = acc[b] + prod;                               // v2.cu:66-70
            }
        }
    }
    uint8_t *dst = out + (size_t)y * rowbytes + xb;
    for (int b = 0; b < 4 && xb + b < rowbytes; b++) dst[b] = (uint8_t)f32_to_u8(acc[b]);
}

hipError_t launch_conv_kxk(const uint8_t *in, uint8_t *out, int w, int h, const float *kk, int K, FrameBatch fb,
                           hipStream_t s) {
    if (w <= 0 || h <= 0 || fb.nframes <= 0) return hipSuccess;
    const int rowbytes = 3 * w;
    hipLaunchKernelGGL(k_conv_kxk, dim3((rowbytes + 255) / 256, (h + 3) / 4, (unsigned)fb.nframes), dim3(256), 0, s, in, out,
                       rowbytes, h, kk, K, fb.stride);
    return hipGetLastError();
}

hipError_t launch_conv3x3(const uint8_t *in, uint8_t *out, int w, int h, const float *k9, bool k9_symmetric,
                          FrameBatch fb, hipStream_t s) {
    if (w <= 0 || h <= 0 || fb.nframes <= 0) return hipSuccess;
    const int rowbytes = 3 * w;
    // (the strip kernel keeps lanes without work at an offset of 2^31 plus up to 40 rows: rows below 2^31 / 40 bytes)
    if (rowbytes % 16 == 0 && aligned16(in) && aligned16(out) && fb.stride % 16 == 0 && rowbytes < (1 << 25)) {
        const unsigned lanes = (unsigned)(rowbytes / 16) * (unsigned)((h + kStripRows - 1) / kStripRows);   // (band, strip) pairs
        const dim3 grid((lanes + 63u) / 64u, 1, (unsigned)fb.nframes);
        if (k9_symmetric)
            hipLaunchKernelGGL((k_conv3x3_strip<true>), grid, dim3(64), 0, s, in, out, rowbytes, h, k9, fb.stride);
        else
            hipLaunchKernelGGL((k_conv3x3_strip<false>), grid, dim3(64), 0, s, in, out, rowbytes, h, k9, fb.stride);
    } else {
        const dim3 grid((w + kConvTW - 1) / kConvTW, (h + kConvRows - 1) / kConvRows, (unsigned)fb.nframes);
        hipLaunchKernelGGL(k_conv3x3_any, grid, dim3(256), 0, s, in, out, w, h, k9, fb.stride);
    }
    return hipGetLastError();
}

// ---- 5x5 median: tests/noise_filter_benchmark/v3.cu:32-90 ----------------------------------------------
// Per channel the 13th smallest of the 5x5 neighbourhood, zeros outside the image.  A workgroup stages
// (16 + 4) rows x (64 + 4) pixels in LDS and every thread selects the median of 25 bytes with the
// 99-exchange selection network of N. Devillard's "opt_med25" (each exchange = v_min_u32 + v_max_u32; the
// network was re-verified here for all 2^25 zero-one inputs before use).  VALU-bound by construction
// (198 instructions per output byte); the reference documents the filter as too slow to ship.
constexpr int kMedTW = 64, kMedTR = 16;

#define MED_X(a, b) { const uint32_t lo_ = min(p[a], p[b]); p[b] = max(p[a], p[b]); p[a] = lo_; }
__device__ __forceinline__ uint32_t median25(uint32_t (&p)[25]) {
    MED_X(0, 1) MED_X(3, 4) MED_X(2, 4) MED_X(2, 3) MED_X(6, 7) MED_X(5, 7) MED_X(5, 6) MED_X(9, 10)
    MED_X(8, 10) MED_X(8, 9) MED_X(12, 13) MED_X(11, 13) MED_X(11, 12) MED_X(15, 16) MED_X(14, 16) MED_X(14, 15)
    MED_X(18, 19) MED_X(17, 19) MED_X(17, 18) MED_X(21, 22) MED_X(20, 22) MED_X(20, 21) MED_X(23, 24) MED_X(2, 5)
    MED_X(3, 6) MED_X(0, 6) MED_X(0, 3) MED_X(4, 7) MED_X(1, 7) MED_X(1, 4) MED_X(11, 14) MED_X(8, 14)
    MED_X(8, 11) MED_X(12, 15) MED_X(9, 15) MED_X(9, 12) MED_X(13, 16) MED_X(10, 16) MED_X(10, 13) MED_X(20, 23)
    MED_X(17, 23) MED_X(17, 20) MED_X(21, 24) MED_X(18, 24) MED_X(18, 21) MED_X(19, 22) MED_X(8, 17) MED_X(9, 18)
    MED_X(0, 18) MED_X(0, 9) MED_X(10, 19) MED_X(1, 19) MED_X(1, 10) MED_X(11, 20) MED_X(2, 20) MED_X(2, 11)
    MED_X(12, 21) MED_X(3, 21) MED_X(3, 12) MED_X(13, 22) MED_X(4, 22) MED_X(4, 13) MED_X(14, 23) MED_X(5, 23)
    MED_X(5, 14) MED_X(15, 24) MED_X(6, 24) MED_X(6, 15) MED_X(7, 16) MED_X(7, 19) MED_X(13, 21) MED_X(15, 23)
    MED_X(7, 13) MED_X(7, 15) MED_X(1, 9) MED_X(3, 11) MED_X(5, 17) MED_X(11, 17) MED_X(9, 17) MED_X(4, 10)
    MED_X(6, 12) MED_X(7, 14) MED_X(4, 6) MED_X(4, 7) MED_X(12, 14) MED_X(10, 14) MED_X(6, 7) MED_X(10, 12)
    MED_X(6, 10) MED_X(6, 17) MED_X(12, 17) MED_X(7, 17) MED_X(7, 10) MED_X(12, 18) MED_X(7, 12) MED_X(10, 18)
    MED_X(12, 20) MED_X(10, 20) MED_X(10, 12)
    return p[12];
}
#undef MED_X

__global__ __launch_bounds__(256) void k_median5x5(const uint8_t *in, uint8_t *out, int w, int h, size_t stride) {
    constexpr int kRowB = (kMedTW + 4) * 3;
    __shared__ uint8_t tile[kMedTR + 4][kRowB + 4];
    in += (size_t)blockIdx.z * stride;
    out += (size_t)blockIdx.z * stride;
    const int x0 = blockIdx.x * kMedTW, y0 = blockIdx.y * kMedTR;
    for (int i = threadIdx.x; i < (kMedTR + 4) * kRowB; i += 256) {
        const int ry = i / kRowB, rb = i - ry * kRowB;
        const int gy = y0 + ry - 2, gxb = (x0 - 2) * 3 + rb;
        uint8_t v = 0;                                   // zeros outside the image, v3.cu:61-69
        if (gy >= 0 && gy < h && gxb >= 0 && gxb < w * 3) v = in[(size_t)gy * w * 3 + gxb];
        tile[ry][rb] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < kMedTR * kMedTW * 3; o += 256) {
        const int ry = o / (kMedTW * 3), cb = o - ry * (kMedTW * 3);
        const int gx = x0 + cb / 3, gy = y0 + ry;
        if (gx >= w || gy >= h) continue;
        uint32_t p[25];
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) p[i * 5 + j] = tile[ry + i][cb + 3 * j];   // v3.cu:79-84
        out[(size_t)gy * w * 3 + (size_t)x0 * 3 + cb] = (uint8_t)median25(p);       // v3.cu:86-88
    }
}

// k_median5x5_strip (rows a multiple of 8 bytes, 8-byte aligned frames; k_median5x5 above is the form for every other
// geometry).  Round 5: the tile kernel read 25 single bytes from LDS per output and ran the 99-exchange network on one
// output per instruction: 35 us per 1080p frame.  This one
//   * computes TWO outputs per instruction: a register holds the same byte position of two row bands of the frame
//     (rows y and y + bandrows) as two 16-bit halves, and every exchange is v_pk_min_u16 + v_pk_max_u16;
//   * sorts every COLUMN of five rows once (18 instructions) and shares it between the five outputs whose windows
//     contain it; what is left per output is the 13th smallest of 25 values in five ascending groups: the program of
//     csrc/median_net.h (tools/median_net/search.py derives it from a sorter by pruning and proves it on all 7776
//     zero-one inputs with sorted columns);
//   * has no LDS, no barriers and no edge code: a lane owns 8 bytes x 2 bands, walks them top to bottom with a ring
//     of five rows in registers, and takes the sorted columns left and right of its own (the windows reach 6 bytes =
//     two pixels either side) from the neighbour lanes with DPP moves.  Lanes 0 and 63 of a wave only supply their
//     columns (a wave stores 62 strips, neighbouring waves overlap by two); rows and strips outside the image are
//     reads beyond the buffer descriptor's records, i.e. zeros (v3.cu:61-69), and stores there are dropped.
typedef _Float16 med_u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t med_u32x2 __attribute__((ext_vector_type(2)));
constexpr int kMedStripLanes = 62;   // strips a wave stores

__device__ __forceinline__ med_u16x2 med_min(med_u16x2 a, med_u16x2 b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ med_u16x2 med_max(med_u16x2 a, med_u16x2 b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ med_u16x2 med_from(uint32_t v) { return __builtin_bit_cast(med_u16x2, v); }
__device__ __forceinline__ uint32_t med_bits(med_u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

#include "median_net.h"
// the 13th smallest of the 25 values of five ascending columns
__device__ __forceinline__ med_u16x2 median_of_sorted_columns(const med_u16x2 (&c0)[5], const med_u16x2 (&c1)[5],
                                                              const med_u16x2 (&c2)[5], const med_u16x2 (&c3)[5],
                                                              const med_u16x2 (&c4)[5]) {
#define MEDNET_IN(col, rank) (col == 0 ? c0[rank] : col == 1 ? c1[rank] : col == 2 ? c2[rank] : col == 3 ? c3[rank] : c4[rank])
#define MEDNET_MIN(d, a, b) const med_u16x2 d = med_min(a, b);
#define MEDNET_MAX(d, a, b) const med_u16x2 d = med_max(a, b);
#define MEDNET_OUT(x) return x;
    MEDNET_BODY
#undef MEDNET_IN
#undef MEDNET_MIN
#undef MEDNET_MAX
#undef MEDNET_OUT
}

#define MED_CE(a, b) { const med_u16x2 lo_ = med_min(s[a], s[b]); s[b] = med_max(s[a], s[b]); s[a] = lo_; }
__device__ __forceinline__ void med_sort5(med_u16x2 (&s)[5]) {   // 9 exchanges
    MED_CE(0, 1) MED_CE(3, 4) MED_CE(2, 4) MED_CE(2, 3) MED_CE(0, 3) MED_CE(0, 2) MED_CE(1, 4) MED_CE(1, 3) MED_CE(1, 2)
}
#undef MED_CE

__global__ __launch_bounds__(64)
void k_median5x5_strip(const uint8_t *in, uint8_t *out, int rowbytes, int h, int bandrows, int npairs, size_t stride) {
    constexpr uint32_t kOut = 0x80000000u;   // beyond every frame (mi355_create: N < 2^31); + a band's rows it does not wrap
    const uint32_t lane = threadIdx.x;
    const uint32_t frame_bytes = (uint32_t)rowbytes * (uint32_t)h;
    const __amdgpu_buffer_rsrc_t src = conv_rsrc(in + (size_t)blockIdx.z * stride, frame_bytes);
    uint8_t *dst_base = out + (size_t)blockIdx.z * stride;
    const uint32_t strips = (uint32_t)rowbytes / 8u;
    // The frame's (pair of bands, strip) places are dealt to the lanes in one running number u, every pair of bands with
    // one empty place in front of its first strip and one behind its last (strips -1 and `strips`: lanes whose loads
    // find no records, i.e. the zero columns left and right of the image).  Lane l of wave w is place 62 w - 1 + l:
    // lanes 1..62 store, lanes 0 and 63 repeat the neighbour waves' places 62 and 1 and only supply their columns.
    const int u = (int)(blockIdx.x * kMedStripLanes + lane) - 1;
    const int per_pair = (int)strips + 2;
    const int pair = u >= 0 ? u / per_pair : 0;
    const int strip = u - pair * per_pair - 1;
    const bool inside = u >= 0 && pair < npairs && strip >= 0 && (uint32_t)strip < strips;
    // band A = rows ya .. ya + bandrows - 1, band B the bandrows rows below it; the walk starts two rows above a band.
    // Offsets are modulo 2^32: a row above the image is a huge offset (no records there: zeros) until the walk reaches row 0.
    const int ya = pair * 2 * bandrows;
    const uint32_t colA = inside ? (uint32_t)((ya - 2) * rowbytes) + (uint32_t)strip * 8u : kOut;
    const uint32_t colB = inside ? colA + (uint32_t)(bandrows * rowbytes) : kOut;
    const bool stores = inside && lane >= 1u && lane <= (uint32_t)kMedStripLanes;
    const uint32_t outA = stores ? colA : kOut, outB = stores ? colB : kOut;

    // The halves of a register are the bytes themselves, 0x0000 .. 0x00ff: as half floats, zero and the first 255 DENORMALS,
    // whose order is the bytes' order.  Minimum and maximum return them unchanged as long as half-float denormals are not
    // flushed -- the compiler's default for this target (.amdhsa_float_denorm_mode_16_64 3), set here regardless: MODE bits
    // 4..7 = keep denormals as inputs and results, single and half / double (the kernel has no other arithmetic).
    __builtin_amdgcn_s_setreg((4 - 1) << 11 | 4 << 6 | 1 /* hwreg(HW_REG_MODE, 4, 4) */, 0xf);
    struct Raw { med_u32x2 a, b; };
    auto load_row = [&](int i) {   // row ya - 2 + i of band A, the same row of band B
        Raw v;
        v.a = __builtin_bit_cast(med_u32x2, __builtin_amdgcn_raw_buffer_load_b64(src, colA + (uint32_t)i * (uint32_t)rowbytes, 0, 0));
        v.b = __builtin_bit_cast(med_u32x2, __builtin_amdgcn_raw_buffer_load_b64(src, colB + (uint32_t)i * (uint32_t)rowbytes, 0, 0));
        return v;
    };
    // byte q of band A in the low half, of band B in the high half
    auto unpack = [&](const Raw &v, med_u16x2 (&p)[8]) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t sel = 0x0c000c00u | (uint32_t)(q & 3) | ((uint32_t)(4 + (q & 3)) << 16);
            p[q] = med_from(__builtin_amdgcn_perm(q < 4 ? v.b.x : v.b.y, q < 4 ? v.a.x : v.a.y, sel));
        }
    };

    med_u16x2 ring[5][8];   // the five rows around the output row, in the order they arrived modulo 5 (a sort does not care)
    Raw raw;
    // One output row: `raw` (row k + 4 of the walk) completes the five rows, row k + 5 is requested, the columns are sorted
    // and the 8 positions x 2 bands are selected and stored.
    auto step = [&](int k, med_u16x2 (&fresh)[8]) {
        unpack(raw, fresh);
        __builtin_amdgcn_sched_barrier(0);
        raw = load_row(k + 5);
        __builtin_amdgcn_sched_barrier(0);
        med_u16x2 s[8][5];   // the sorted columns of the lane's own positions 0 .. 7
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int r = 0; r < 5; r++) s[q][r] = ring[r][q];
            med_sort5(s[q]);
        }
        uint32_t m[8];
        // v3.cu:79-88: the window of position q is the columns q - 6, q - 3, q, q + 3, q + 6 (one colour channel).  The
        // three channels in turn (fewer columns alive at once): positions c - 6, c - 3, .. <= 13, of which the ones below 0
        // are the left neighbour's positions + 8 and the ones above 7 the right neighbour's positions - 8.
#pragma unroll
        for (int c = 0; c < 3; c++) {
            med_u16x2 col[7][5];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const int pos = c - 6 + 3 * j;
                if (pos > 13) continue;
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    if (pos < 0)
                        col[j][r] = med_from((uint32_t)__builtin_amdgcn_update_dpp(0, (int)med_bits(s[pos + 8 < 8 ? pos + 8 : 0][r]), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
                    else if (pos > 7)
                        col[j][r] = med_from((uint32_t)__builtin_amdgcn_update_dpp(0, (int)med_bits(s[pos - 8 >= 0 ? pos - 8 : 0][r]), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
                    else
                        col[j][r] = s[pos][r];
                }
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int q = c + 3 * i;
                if (q > 7) continue;
                // the channel's windows col[0..4], col[1..5], col[2..6] as (what they have in common; their own two columns):
                // the selection program reads its first three columns first, and what it computes from them alone is the
                // same expression in all three calls -- computed once (tools/median_net/anneal.cpp, SHARE)
                m[q] = med_bits(median_of_sorted_columns(col[2], col[3], col[4], col[i == 2 ? 5 : i], col[i == 0 ? 1 : i + 4]));
                // one output at a time (the registers of one selection program, not of three): the empty statement makes the
                // result exist HERE -- a value nobody reads until the row is stored is otherwise placed behind every barrier
                asm volatile("" : "+v"(m[q]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // m[q] = (band A's byte q, 0, band B's byte q, 0)
        const uint32_t t01 = __builtin_amdgcn_perm(m[1], m[0], 0x06020400u), t23 = __builtin_amdgcn_perm(m[3], m[2], 0x06020400u);
        const uint32_t t45 = __builtin_amdgcn_perm(m[5], m[4], 0x06020400u), t67 = __builtin_amdgcn_perm(m[7], m[6], 0x06020400u);
        const med_u32x2 oa = {__builtin_amdgcn_perm(t23, t01, 0x05040100u), __builtin_amdgcn_perm(t67, t45, 0x05040100u)};
        const med_u32x2 ob = {__builtin_amdgcn_perm(t23, t01, 0x07060302u), __builtin_amdgcn_perm(t67, t45, 0x07060302u)};
        // (no branch around the stores of the steps that round a band up to whole groups of five: a descriptor without
        // records drops them; rows below the image are beyond the records by themselves)
        const __amdgpu_buffer_rsrc_t d = conv_rsrc(dst_base, k < bandrows ? frame_bytes : 0u);
        const uint32_t rowoff = (uint32_t)(k + 2) * (uint32_t)rowbytes;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(med_u32x2, oa), d, outA + rowoff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(med_u32x2, ob), d, outB + rowoff, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int i = 0; i < 4; i++) {
        raw = load_row(i);
        unpack(raw, ring[i]);
    }
    raw = load_row(4);
    for (int k = 0; k < bandrows; k += 5) {
        step(k, ring[4]);
        step(k + 1, ring[0]);
        step(k + 2, ring[1]);
        step(k + 3, ring[2]);
        step(k + 4, ring[3]);
    }
}

// rows per band of the strip kernel: whole groups of five.  A wave walks a PAIR of bands, so the rows walked per strip are
// pairs x rows >= h / 2: the band length that wastes least of the last pair wins (1080 rows: 20, 30, 45 or 60 -- 27, 18, 12
// or 9 full pairs; 40 would walk 14 x 40 = 560 rows for 540), and among equals the SHORTEST from 20 rows up: they measure
// the same (a band reads four rows more than it writes, but HBM is idle; profiles/r05y_median_rows.txt) and more, shorter
// waves leave a shorter tail in the chip's last round.  Fewer than 4096 waves: shorter bands still (one frame: 5 rows, 1260
// waves at 1080p).
static int median_bandrows(int h, int strips, int nframes) {
    auto waves = [&](int rows) {
        const long pairs = (h + 2 * rows - 1) / (2 * rows);
        return (pairs * (strips + 2) + kMedStripLanes - 1) / kMedStripLanes * nframes;
    };
    int best = 20;
    long best_rows = -1;
    for (int rows = 20; rows <= 60; rows += 5) {
        const long walked = (long)((h + 2 * rows - 1) / (2 * rows)) * rows;
        if (best_rows < 0 || walked < best_rows) { best = rows; best_rows = walked; }
    }
    while (best > 5 && waves(best) < 4096) best -= 5;
    return best;
}

hipError_t launch_median5x5(const uint8_t *in, uint8_t *out, int w, int h, int rows_per_band, FrameBatch fb, hipStream_t s) {
    if (w <= 0 || h <= 0 || fb.nframes <= 0) return hipSuccess;
    const size_t rowbytes = (size_t)w * 3;
    if (rowbytes % 8 == 0 && ((uintptr_t)in & 7u) == 0 && ((uintptr_t)out & 7u) == 0 && fb.stride % 8 == 0 && rowbytes < (1u << 24)) {
        const int strips = (int)(rowbytes / 8);
        const int bandrows = rows_per_band > 0 ? rows_per_band : median_bandrows(h, strips, fb.nframes);
        const int pairs = (h + 2 * bandrows - 1) / (2 * bandrows);
        const dim3 grid((unsigned)(((long)pairs * (strips + 2) + kMedStripLanes - 1) / kMedStripLanes), 1, (unsigned)fb.nframes);
        hipLaunchKernelGGL(k_median5x5_strip, grid, dim3(64), 0, s, in, out, (int)rowbytes, h, bandrows, pairs, fb.stride);
        return hipGetLastError();
    }
    const dim3 grid((w + kMedTW - 1) / kMedTW, (h + kMedTR - 1) / kMedTR, (unsigned)fb.nframes);
    hipLaunchKernelGGL(k_median5x5, grid, dim3(256), 0, s, in, out, w, h, fb.stride);
    return hipGetLastError();
}

// ---- text overlay: kernel2_char, kernels.cu:351-375 (row-exact blit, no 32-byte straddle) ---------
__global__ __launch_bounds__(256) void k_blit_glyph(uint8_t *frame, const uint8_t *glyph, int glyph_h,
                                                    int glyph_wbytes, int x_off_bytes, int frame_wbytes,
                                                    int frame_h) {
    const int total = glyph_h * glyph_wbytes;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / glyph_wbytes, x = x_off_bytes + i % glyph_wbytes;   // kernels.cu:344-346
        if (y < frame_h && x < frame_wbytes) frame[(size_t)y * frame_wbytes + x] = glyph[i];
    }
}

hipError_t launch_blit_glyph(uint8_t *frame, const uint8_t *glyph, int glyph_h, int glyph_wbytes,
                             int x_off_bytes, int frame_wbytes, int frame_h, hipStream_t s) {
    const int total = glyph_h * glyph_wbytes;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_blit_glyph, dim3((total + 255) / 256), dim3(256), 0, s, frame, glyph, glyph_h,
                       glyph_wbytes, x_off_bytes, frame_wbytes, frame_h);
    return hipGetLastError();
}

}  // namespace mi355
