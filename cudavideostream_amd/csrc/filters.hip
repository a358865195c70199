// filters.hip -- the filter kernels that sit on the same frame as the diff (gfx950).
//
// Each kernel cites the reference kernel it replaces (server/src/kernels.cu) and the CPU statement
// whose results it reproduces bit for bit.  All are byte-streaming, HBM-bound kernels: 16 pixels
// (48 B = three 16-B loads) per lane, grid-stride free (one pass), no atomics except the histogram.
// This file is compiled with -ffp-contract=off: the weighted grayscale and the convolution depend on
// separate multiply and add roundings.
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
template <bool WEIGHTED>
__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r) {
    if (WEIGHTED) {
        const double v = 0.114 * (double)b + 0.587 * (double)g + 0.299 * (double)r;
        return (uint32_t)v;
    }
    return (b + g + r) / 3u;
}

template <bool WEIGHTED, bool FAST>
__global__ __launch_bounds__(256) void k_gray(const uint8_t *in, uint8_t *out, uint32_t npix) {
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    if (FAST && rem >= 16) {
        const Px16 p = load_px16<true>(in + off, 48);
        Px16 q;
#pragma unroll
        for (int i = 0; i < 12; i++) q.w[i] = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t v =
                gray_of<WEIGHTED>(get_byte(p, 3 * k), get_byte(p, 3 * k + 1), get_byte(p, 3 * k + 2));
            put_byte(q, 3 * k, v); put_byte(q, 3 * k + 1, v); put_byte(q, 3 * k + 2, v);
        }
        store_px16<true>(out + off, q, 48);
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

hipError_t launch_gray(const uint8_t *in, uint8_t *out, uint32_t npix, bool weighted, hipStream_t s) {
    if (npix == 0) return hipSuccess;
    const bool fast = aligned16(in) && aligned16(out);
    const dim3 g = px16_grid(npix), b(256);
    if (weighted) {
        if (fast) hipLaunchKernelGGL((k_gray<true, true>), g, b, 0, s, in, out, npix);
        else hipLaunchKernelGGL((k_gray<true, false>), g, b, 0, s, in, out, npix);
    } else {
        if (fast) hipLaunchKernelGGL((k_gray<false, true>), g, b, 0, s, in, out, npix);
        else hipLaunchKernelGGL((k_gray<false, false>), g, b, 0, s, in, out, npix);
    }
    return hipGetLastError();
}

// ---- binarize chain: kernels.cu:138-241, CPU semantics server/src/server.cpp:103-135 -----------------
// Histogram of every 3rd byte (one sample per pixel): per-wave private LDS bins, merged per workgroup,
// then 256 global atomics per workgroup (integer adds: order-independent, deterministic).
__global__ __launch_bounds__(256) void k_histogram(const uint8_t *gray3, uint32_t npix, int32_t *hist,
                                                   uint32_t px_per_block) {
    __shared__ int32_t bins[4][256];
    for (int i = threadIdx.x; i < 1024; i += 256) (&bins[0][0])[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const uint32_t p0 = blockIdx.x * px_per_block;
    const uint32_t p1 = min(npix, p0 + px_per_block);
    for (uint32_t p = p0 + threadIdx.x; p < p1; p += 256)
        atomicAdd(&bins[wave][gray3[(size_t)p * 3]], 1);
    __syncthreads();
    const int v = bins[0][threadIdx.x] + bins[1][threadIdx.x] + bins[2][threadIdx.x] +
                  bins[3][threadIdx.x];
    if (v) atomicAdd(&hist[threadIdx.x], v);
}

// server.cpp:108-127 executed as written by one lane (256 iterations; the dead `else if` included).
__global__ void k_two_max_threshold(const int32_t *histogram, int32_t *thr_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int max = -1, sec_max = -1;
    int index_max = -1, index_sec_max = -1;
    for (int i = 0; i < 256; i++) {
        const int h = histogram[i];
        if (h >= max) {
            index_sec_max = index_max;
            index_max = i;
            max = h;
            sec_max = max;
        } else if (h > sec_max && h < max) {
            sec_max = h;
            index_sec_max = i;
        }
    }
    int threshold = (index_max + index_sec_max) / 2;
    if (threshold < 50) threshold = 50;
    if (threshold > 200) threshold = 200;
    *thr_out = threshold;
}

// kernels.cu:222-241 / server.cpp:129-135: byte > thr ? 255 : 0, 16 bytes per lane.
__global__ __launch_bounds__(256) void k_binarize(const uint8_t *in, uint8_t *out, uint32_t nbytes,
                                                  const int32_t *thr_p, bool fast) {
    const uint32_t thr = (uint32_t)*thr_p;
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

hipError_t launch_binarize_chain(const uint8_t *gray, uint8_t *out, uint32_t nbytes, int32_t *hist,
                                 int32_t *thr, hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist, 0, 256 * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    const uint32_t npix = nbytes / 3;
    if (npix) {
        const uint32_t px_per_block = 256 * 32;
        hipLaunchKernelGGL(k_histogram, dim3((npix + px_per_block - 1) / px_per_block), dim3(256), 0, s,
                           gray, npix, hist, px_per_block);
    }
    hipLaunchKernelGGL(k_two_max_threshold, dim3(1), dim3(64), 0, s, hist, thr);
    if (nbytes) {
        const uint32_t lanes = (nbytes + 15) / 16;
        hipLaunchKernelGGL(k_binarize, dim3((lanes + 255) / 256), dim3(256), 0, s, gray, out, nbytes, thr,
                           aligned16(gray) && aligned16(out));
    }
    return hipGetLastError();
}

// ---- heat map: kernels.cu:243-270, CPU tests/heat_map_benchmark/cpu.cu:19-27,54-66 -----------------
// d = |dB|+|dG|+|dR| (0..765) -> 766-entry BGR look-up table built on the host with the reference's
// exact double expression; staged in LDS.
template <bool FAST>
__global__ __launch_bounds__(256) void k_heat_map(const uint8_t *cur, const uint8_t *prev, uint8_t *out,
                                                  uint32_t npix, const uint8_t *lut) {
    __shared__ uint8_t s_lut[768 * 3];
    for (int i = threadIdx.x; i < 766 * 3; i += 256) s_lut[i] = lut[i];
    __syncthreads();
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    const bool full = FAST && rem >= 16;
    const int nb = full ? 48 : (int)(rem < 16 ? rem : 16) * 3;
    const Px16 c = full ? load_px16<true>(cur + off, 48) : load_px16<false>(cur + off, nb);
    const Px16 p = full ? load_px16<true>(prev + off, 48) : load_px16<false>(prev + off, nb);
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
    if (full) store_px16<true>(out + off, q, 48);
    else store_px16<false>(out + off, q, nb);
}

hipError_t launch_heat_map(const uint8_t *cur, const uint8_t *prev, uint8_t *out, uint32_t npix,
                           const uint8_t *lut, hipStream_t s) {
    if (npix == 0) return hipSuccess;
    const bool fast = aligned16(cur) && aligned16(prev) && aligned16(out);
    if (fast) hipLaunchKernelGGL((k_heat_map<true>), px16_grid(npix), dim3(256), 0, s, cur, prev, out, npix, lut);
    else hipLaunchKernelGGL((k_heat_map<false>), px16_grid(npix), dim3(256), 0, s, cur, prev, out, npix, lut);
    return hipGetLastError();
}

// ---- red motion map, dense: tests/heat_map_red_benchmark/cpu.cu:38-55 (test.cu:142-168) -----------
template <bool FAST>
__global__ __launch_bounds__(256) void k_red_dense(const uint8_t *cur, const uint8_t *prev, uint8_t *out,
                                                   uint32_t npix, int thr) {
    const uint32_t lane_px = (blockIdx.x * 256u + threadIdx.x) * 16u;
    if (lane_px >= npix) return;
    const uint32_t rem = npix - lane_px;
    const size_t off = (size_t)lane_px * 3;
    const bool full = FAST && rem >= 16;
    const int nb = full ? 48 : (int)(rem < 16 ? rem : 16) * 3;
    const Px16 c = full ? load_px16<true>(cur + off, 48) : load_px16<false>(cur + off, nb);
    const Px16 p = full ? load_px16<true>(prev + off, 48) : load_px16<false>(prev + off, nb);
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
    if (full) store_px16<true>(out + off, q, 48);
    else store_px16<false>(out + off, q, nb);
}

hipError_t launch_red_dense(const uint8_t *cur, const uint8_t *prev, uint8_t *out, uint32_t npix,
                            int thr, hipStream_t s) {
    if (npix == 0) return hipSuccess;
    const bool fast = aligned16(cur) && aligned16(prev) && aligned16(out);
    if (fast) hipLaunchKernelGGL((k_red_dense<true>), px16_grid(npix), dim3(256), 0, s, cur, prev, out, npix, thr);
    else hipLaunchKernelGGL((k_red_dense<false>), px16_grid(npix), dim3(256), 0, s, cur, prev, out, npix, thr);
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

// ---- 3x3 noise filter: kernels.cu:97-136 --------------------------------------------------------------
// out[y][x][c] = (uint8) sum_{i,j} k[3i+j] * in[y+i-1][x+j-1][c], zero outside the image, float
// accumulator, taps in i-major / j-minor order, one multiply then one add per tap.
// A workgroup stages ROWS+2 input rows of a 3*TW-byte column band (+1 pixel halo each side) in LDS.
constexpr int kConvTW = 64;    // output pixels per band
constexpr int kConvRows = 8;   // output rows per workgroup

__global__ __launch_bounds__(256) void k_conv3x3(const uint8_t *in, uint8_t *out, int w, int h,
                                                 const float *k9) {
    __shared__ uint8_t tile[kConvRows + 2][(kConvTW + 2) * 3 + 2];
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
        out[((size_t)gy * w) * 3 + (size_t)x0 * 3 + cb] = (uint8_t)acc;   // kernels.cu:131-133
    }
}

hipError_t launch_conv3x3(const uint8_t *in, uint8_t *out, int w, int h, const float *k9, hipStream_t s) {
    if (w <= 0 || h <= 0) return hipSuccess;
    const dim3 grid((w + kConvTW - 1) / kConvTW, (h + kConvRows - 1) / kConvRows);
    hipLaunchKernelGGL(k_conv3x3, grid, dim3(256), 0, s, in, out, w, h, k9);
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
