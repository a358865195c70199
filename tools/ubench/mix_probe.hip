// What tells a box with the slow dense expansion (S0: 265 us per 32 pairs) from one with the fast (205)?  Plain reads, plain
// writes and the clock do not (bench.py's `board`).  Mixed traffic patterns, each a grid-stride kernel over 1 GiB:
//   copy      16 B per lane read + 16 B written (1 : 1)
//   read4     four 16-B reads per 16-B write
//   narrow    a 4-B index + a 1-B value written per lane while 1 B + 4 B per lane are read (the expansion's shape)
//   gather    16-B reads at a pseudo-random 64-KiB-page order (row misses) + linear 16-B writes
//   hipcc -O3 --offload-arch=gfx950 -o mix_probe mix_probe.hip && ./mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const u32x4 *in, u32x4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
__global__ __launch_bounds__(256) void k_read4(const u32x4 *in, u32x4 *out, size_t n) {   // n = vectors of out; in has 4 n
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 a = __builtin_nontemporal_load(in + i), b = __builtin_nontemporal_load(in + n + i);
        const u32x4 c = __builtin_nontemporal_load(in + 2 * n + i), d = __builtin_nontemporal_load(in + 3 * n + i);
        __builtin_nontemporal_store(a ^ b ^ c ^ d, out + i);
    }
}
__global__ __launch_bounds__(256) void k_narrow(const uint8_t *rec, const uint32_t *codes, uint32_t *xs, uint8_t *df, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t c = __builtin_nontemporal_load(codes + i);
        const uint8_t r = __builtin_nontemporal_load(rec + i);
        __builtin_nontemporal_store(c + (uint32_t)i, xs + i);
        __builtin_nontemporal_store((uint8_t)(r + 1), df + i);
    }
}
__global__ __launch_bounds__(256) void k_gather(const u32x4 *in, u32x4 *out, size_t n) {   // n a power of two (vectors)
    const size_t pages = n >> 12;   // 4096 vectors = 64 KiB
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t page = i >> 12, within = i & 4095;
        const size_t src = ((page * 2654435761ull) & (pages - 1)) << 12 | within;
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + src), out + i);
    }
}

template <typename F>
static void timeit(const char *name, double bytes, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-8s %8.1f GB/s (%.3f ms, %.2f GB moved)\n", name, bytes / (best * 1e-3) / 1e9, best, bytes / 1e9);
}

int main() {
    const size_t GiB = 1ull << 30;
    uint8_t *a, *b;
    CK(hipMalloc((void **)&a, 4 * GiB)); CK(hipMalloc((void **)&b, 2 * GiB));
    CK(hipMemset(a, 0x5a, 4 * GiB)); CK(hipMemset(b, 0x33, 2 * GiB));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const unsigned blocks = prop.multiProcessorCount * 8;
    const size_t nv = GiB / 16;
    timeit("copy", 2.0 * GiB, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)a, (u32x4 *)b, nv); });
    timeit("read4", 5.0 * GiB, [&] { hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)a, (u32x4 *)b, nv); });
    const size_t ne = GiB / 5;   // entries: 1 + 4 bytes read, 4 + 1 written
    timeit("narrow", 10.0 * ne, [&] { hipLaunchKernelGGL(k_narrow, dim3(blocks), dim3(256), 0, 0, a, (const uint32_t *)(a + GiB), (uint32_t *)b, b + GiB + GiB / 2, ne); });
    timeit("gather", 2.0 * GiB, [&] { hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, (const u32x4 *)a, (u32x4 *)b, nv); });
    return 0;
}
