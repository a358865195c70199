// stream_ops.hip -- the two small operators either side of the packed stream:
//
//   k_apply / k_apply_all : the client's reconstruction, client/opencv.cpp:64-66
//                           (`frame2.data[xs[i]] += buffer[i]` for the pos entries of a frame);
//   k_merge_parts         : concatenation of the streams of the row bands of ONE video stream that
//                           several cores (GPUs) packed independently into
//                           the single stream the sender would have produced.
//
// Both are index-driven byte scatter/copy: HBM-latency work with 5 bytes of traffic per entry, no
// arithmetic worth naming.
#include "internal.h"

namespace mi355 {

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// One frame.  Indices of a frame are distinct (strictly ascending, tests/cuda_streaming/test.cu:563-573),
// so plain byte read-modify-writes do not race.  count comes from the device (offsets) or the host.
// xs is read with byte alignment because the wire format puts it at any address.
__global__ __launch_bounds__(256) void k_apply(uint8_t *frame, uint32_t nbytes, const uint8_t *xs,
                                               const uint8_t *diff, const uint32_t *d_offsets, int t,
                                               uint32_t host_count) {
    uint32_t first = 0, count = host_count;
    if (d_offsets) {
        first = d_offsets[t];
        count = d_offsets[t + 1] - first;
    }
    const uint32_t step = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += step) {
        const uint32_t x = load_u32_unaligned(xs + 4 * (size_t)(first + i));
        if (x < nbytes) frame[x] = (uint8_t)(frame[x] + diff[first + i]);   // opencv.cpp:65
    }
}

// All frames of a batch at once, final frame only: per-byte addition modulo 256 commutes, so entries of
// different frames may land in any order as long as each add is atomic on its byte -- a 32-bit
// compare-and-swap on the containing dword (the frame buffer is dword padded by the allocator).
__global__ __launch_bounds__(256) void k_apply_all(uint8_t *frame, uint32_t nbytes, const int32_t *xs,
                                                   const uint8_t *diff, const uint32_t *d_offsets,
                                                   int nframes) {
    const uint32_t count = d_offsets[nframes];
    const uint32_t step = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += step) {
        const uint32_t x = (uint32_t)xs[i];
        if (x >= nbytes) continue;
        const uint32_t d = diff[i];
        uint32_t *p = (uint32_t *)(frame + (x & ~3u));
        const uint32_t sh = 8u * (x & 3u);
        uint32_t old = *p, assumed;
        do {
            assumed = old;
            const uint32_t byte = ((assumed >> sh) + d) & 0xffu;
            old = atomicCAS(p, assumed, (assumed & ~(0xffu << sh)) | (byte << sh));
        } while (old != assumed);
    }
}

hipError_t launch_apply(uint8_t *frame, uint32_t nbytes, const void *xs, const void *diff,
                        const uint32_t *d_offsets, int t, uint32_t host_count, hipStream_t s) {
    uint32_t blocks = 512;   // device-side counts: a fixed grid strides over whatever the frame holds
    if (!d_offsets) {
        if (host_count == 0) return hipSuccess;
        blocks = (host_count + 255u) / 256u;
        if (blocks > 2048u) blocks = 2048u;
    }
    hipLaunchKernelGGL(k_apply, dim3(blocks), dim3(256), 0, s, frame, nbytes, (const uint8_t *)xs,
                       (const uint8_t *)diff, d_offsets, t, host_count);
    return hipGetLastError();
}

hipError_t launch_apply_all(uint8_t *frame, uint32_t nbytes, const int32_t *xs, const uint8_t *diff,
                            const uint32_t *d_offsets, int nframes, hipStream_t s) {
    hipLaunchKernelGGL(k_apply_all, dim3(4096), dim3(256), 0, s, frame, nbytes, xs, diff, d_offsets, nframes);
    return hipGetLastError();
}

// ---- export of one packed frame into host-mapped (pinned) buffers ---------------------------------------
// The pipelined per-frame path has no host synchronisation between the pack and the copies back
// (the reference reads the count, synchronises, then sizes two cudaMemcpy with it,
// server/src/kernels.cu:507-524): the count stays on the device and this kernel stores exactly `count`
// entries, and the count itself, through the PCIe-mapped pointers.
__global__ __launch_bounds__(256) void k_export(const uint32_t *offsets, const int32_t *xs, const uint8_t *diff,
                                                int32_t *h_xs, uint8_t *h_diff, uint32_t *h_count) {
    const uint32_t count = offsets[1] - offsets[0];
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
    if (gid == 0) *h_count = count;
    // 16-byte stores where the host pointers allow (the device arrays are allocation aligned)
    if (((uintptr_t)h_xs & 15u) == 0) {
        const uint32_t q = count / 4;
        for (uint32_t i = gid; i < q; i += step) ((uint4 *)h_xs)[i] = ((const uint4 *)xs)[i];
        for (uint32_t i = 4 * q + gid; i < count; i += step) h_xs[i] = xs[i];
    } else {
        for (uint32_t i = gid; i < count; i += step) h_xs[i] = xs[i];
    }
    if (((uintptr_t)h_diff & 15u) == 0) {
        const uint32_t q = count / 16;
        for (uint32_t i = gid; i < q; i += step) ((uint4 *)h_diff)[i] = ((const uint4 *)diff)[i];
        for (uint32_t i = 16 * q + gid; i < count; i += step) h_diff[i] = diff[i];
    } else {
        for (uint32_t i = gid; i < count; i += step) h_diff[i] = diff[i];
    }
}

hipError_t launch_export(const uint32_t *offsets, const int32_t *xs, const uint8_t *diff, int32_t *h_xs,
                         uint8_t *h_diff, uint32_t *h_count, hipStream_t s) {
    hipLaunchKernelGGL(k_export, dim3(256), dim3(256), 0, s, offsets, xs, diff, h_xs, h_diff, h_count);
    return hipGetLastError();
}

// ---- merge of row-band streams -------------------------------------------------------------------------
// Part p (a row band, bands ordered top to bottom) holds its own packed stream of the same T frames:
// index part_off[p][0..T], entries at xs_all/diff_all[part_base[p] + ...], byte indices relative to
// the band.  Frame t of the merged stream is the parts' frame-t segments in part order with
// xs + xs_bias[p]: the ascending order of tests/cuda_streaming/test.cu:563-573 over the whole frame.
//   k_merge_index : out_offsets[t] = sum_p part_off[p][t]           (grid-stride over t)
//   k_merge_parts : workgroup (t, p) copies its segment              (grid = (T, nparts))
__global__ __launch_bounds__(256) void k_merge_index(const uint32_t *part_off, int nparts, int nframes,
                                                     uint32_t *out_offsets) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= nframes; t += gridDim.x * blockDim.x) {
        uint32_t acc = 0;
        for (int p = 0; p < nparts; p++) acc += part_off[(size_t)p * (nframes + 1) + t];
        out_offsets[t] = acc;
    }
}

__global__ __launch_bounds__(256) void k_merge_parts(const MergeArgs a) {
    const int t = blockIdx.x, p = blockIdx.y;
    const uint32_t *po = a.part_off + (size_t)p * (a.nframes + 1);
    const uint32_t src0 = po[t], cnt = po[t + 1] - src0;
    uint32_t dst = 0;   // entries of all parts in frames < t, plus parts < p in frame t
    for (int q = 0; q < a.nparts; q++) {
        const uint32_t *qo = a.part_off + (size_t)q * (a.nframes + 1);
        dst += q < p ? qo[t + 1] : qo[t];
    }
    const size_t src = (size_t)a.part_base[p] + src0;
    const int32_t bias = a.xs_bias[p];
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
        if ((size_t)dst + i < a.capacity) {
            a.out_xs[dst + i] = a.xs_all[src + i] + bias;
            a.out_diff[dst + i] = a.diff_all[src + i];
        }
    }
}

hipError_t launch_merge(const MergeArgs &a, uint32_t *out_offsets, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_index, dim3((a.nframes + 256) / 256), dim3(256), 0, s, a.part_off, a.nparts,
                       a.nframes, out_offsets);
    if (a.nframes > 0 && a.nparts > 0)
        hipLaunchKernelGGL(k_merge_parts, dim3(a.nframes, a.nparts), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace mi355
