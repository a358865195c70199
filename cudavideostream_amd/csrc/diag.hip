// diag.hip -- diagnostics behind the C-ABI (include/mi355diff.h, "measurement"): the shader clock the chip holds
// under an integer-VALU load.
//
// Why: the diff/threshold/pack path is bound by instruction issue, so its time follows the
// shader clock, and MI355X boards hold different clocks under the same load (the chip lowers its clock under load;
// devices differ by up to ~12 %, MI355X_MICROARCH.md "DVFS give-back").  bench.py prints this figure next to its
// frames/s so that a box-to-box spread of the headline can be told from a change of the code.
//
// Method (the guide's item 6): stamp s_memtime (shader cycles) and s_memrealtime (a constant 100 MHz counter) around
// a loop of plain integer instructions in every wave; clock = d(memtime) / d(memrealtime) x 100 MHz, median over the
// waves.  The stamps go to a buffer of their own that nothing else reads; the kernel is not part of any product path.
#include <algorithm>
#include <vector>

#include "../../include/mi355diff.h"
#include "internal.h"

namespace mi355 {

__global__ __launch_bounds__(256) void k_clock_probe(uint64_t *stamps, uint32_t iters, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x + 17u, c = a ^ 0x9e3779b9u, d = b + a;
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {   // four independent chains of the pack kernel's instruction classes
            a = (a | 0x80808080u) - (b & 0x7f7f7f7fu);
            b = __builtin_amdgcn_bitop3_b32(b, c, d, 0xb2);
            c = (c + a) ^ (d >> 3);
            d = __builtin_amdgcn_perm(d, a, 0x07020500u) + b;
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
    if ((a ^ b ^ c ^ d) == 0x12345u && iters == 0xffffffffu) stamps[0] = a;   // keeps the chains alive
}

// Streaming read of a buffer, 16 bytes per lane and instruction, four loads in flight per wave: what this board's
// memory system delivers to a read-only kernel (the pack kernel's time follows it: boards differ by ~7 %).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_hbm_read_probe(const u32x4 *buf, size_t nvec, uint32_t *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(buf + i), b = __builtin_nontemporal_load(buf + i + stride);
        const u32x4 c = __builtin_nontemporal_load(buf + i + 2 * stride), d = __builtin_nontemporal_load(buf + i + 3 * stride);
        acc += (a.x ^ b.y) + (c.z ^ d.w);
    }
    for (; i < nvec; i += stride) acc += __builtin_nontemporal_load(buf + i).x;
    if (acc == 0x12345679u) sink[0] = acc;
}

// Streaming writes: WIDE = 16 bytes per lane and instruction (whole lines, like the filters' outputs); NARROW = what the
// dense expansion emits per entry: a 4-byte index and a 1-byte value to two arrays, a lane per entry (256 + 64
// contiguous bytes per wave instruction), non-temporal like the expander's.  Boards whose plain read is the same differ
// by a quarter here (the S0 regime follows it).
template <bool NARROW>
__global__ __launch_bounds__(256) void k_hbm_write_probe(u32x4 *buf, size_t nvec, uint32_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (NARROW) {
        // entries: nvec * 16 / 5; indices in the first 4/5 of the buffer, values behind them
        const size_t entries = nvec * 16 / 5 / 4 * 4;
        uint32_t *xs = reinterpret_cast<uint32_t *>(buf);
        uint8_t *df = reinterpret_cast<uint8_t *>(buf) + entries * 4;
        for (size_t i = i0; i < entries; i += stride) {
            __builtin_nontemporal_store((uint32_t)i ^ seed, xs + i);
            __builtin_nontemporal_store((uint8_t)(i + seed), df + i);
        }
    } else {
        for (size_t i = i0; i < nvec; i += stride) {
            const u32x4 v = {(uint32_t)i, seed, (uint32_t)i ^ seed, 0x5a5a5a5au};
            __builtin_nontemporal_store(v, buf + i);
        }
    }
}

}  // namespace mi355

using namespace mi355;

// One probe: a temporary buffer on the core's device, the kernel four times on the core's stream, the best of the last
// three passes.  `launch(buffer, bytes, stream)` starts the kernel; returns GB/s of `bytes_counted`.
template <typename Launch>
static int run_hbm_probe(mi355_core *c, size_t megabytes, double *gbps, const char *what, Launch launch) {
    if (!c || !gbps) return set_error(MI355_ERR_INVALID, "null argument");
    if (megabytes < 64 || megabytes > 16384) return set_error(MI355_ERR_INVALID, "megabytes outside [64, 16384]");
    // the probe's buffers must live on the core's device, whatever device the calling thread used last (a process that
    // holds cores on several GPUs: a group's members)
    if (hipSetDevice(core_device(c)) != hipSuccess) return set_error(MI355_ERR_HIP, "hipSetDevice");
    hipStream_t s = core_stream(c);
    const size_t bytes = megabytes << 20;
    u32x4 *d = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc((void **)&d, bytes) != hipSuccess) return set_error(MI355_ERR_HIP, "hipMalloc(probe buffer)");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **)&sink, 64);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0x5a, bytes, s);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    hipDeviceProp_t prop{};
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, core_device(c));
    if (e == hipSuccess && prop.multiProcessorCount <= 0) e = hipErrorInvalidDevice;
    float best = 1e30f;
    for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {   // the first pass warms up; the best of the rest counts
        e = hipEventRecord(e0, s);
        if (e == hipSuccess) {
            launch(d, bytes, sink, (unsigned)prop.multiProcessorCount * 8u, s);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return set_error(MI355_ERR_HIP, what);
    *gbps = (double)bytes / (best * 1e-3) / 1e9;
    return MI355_OK;
}

extern "C" int mi355_probe_hbm_read(mi355_core *c, size_t megabytes, double *gbps) {
    return run_hbm_probe(c, megabytes, gbps, "hbm read probe", [](u32x4 *d, size_t bytes, uint32_t *sink, unsigned blocks, hipStream_t s) {
        hipLaunchKernelGGL(k_hbm_read_probe, dim3(blocks), dim3(256), 0, s, d, bytes / 16, sink);
    });
}

extern "C" int mi355_probe_hbm_write(mi355_core *c, size_t megabytes, int narrow, double *gbps) {
    if (narrow != 0 && narrow != 1) return set_error(MI355_ERR_INVALID, "narrow: 0 or 1");
    return run_hbm_probe(c, megabytes, gbps, "hbm write probe", [narrow](u32x4 *d, size_t bytes, uint32_t *, unsigned blocks, hipStream_t s) {
        if (narrow) hipLaunchKernelGGL((k_hbm_write_probe<true>), dim3(blocks), dim3(256), 0, s, d, bytes / 16, 7u);
        else hipLaunchKernelGGL((k_hbm_write_probe<false>), dim3(blocks), dim3(256), 0, s, d, bytes / 16, 7u);
    });
}

extern "C" int mi355_probe_clock(mi355_core *c, int milliseconds, double *shader_mhz) {
    if (!c || !shader_mhz) return set_error(MI355_ERR_INVALID, "null argument");
    if (milliseconds < 1 || milliseconds > 2000) return set_error(MI355_ERR_INVALID, "milliseconds outside [1, 2000]");
    if (hipSetDevice(core_device(c)) != hipSuccess) return set_error(MI355_ERR_HIP, "hipSetDevice");   // see mi355_probe_hbm_read
    hipStream_t s = core_stream(c);
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, core_device(c)) != hipSuccess || prop.multiProcessorCount <= 0)
        return set_error(MI355_ERR_HIP, "hipGetDeviceProperties");
    const int blocks = prop.multiProcessorCount * 4;   // 4 waves per SIMD
    const size_t waves = (size_t)blocks * 4;
    uint64_t *d = nullptr;
    if (hipMalloc((void **)&d, waves * 16) != hipSuccess) return set_error(MI355_ERR_HIP, "hipMalloc(stamps)");
    // 64 instructions per iteration and wave, 4 waves per SIMD, ~2.5-4.5 cycles each: ~800 cycles per iteration
    const uint32_t iters = (uint32_t)((double)milliseconds * 2.0e6 / 800.0);
    std::vector<uint64_t> h(waves * 2);
    hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, s, d, iters, 1u);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, waves * 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(MI355_ERR_HIP, "clock probe");
    std::vector<double> mhz;
    mhz.reserve(waves);
    for (size_t w = 0; w < waves; w++)
        if (h[2 * w + 1]) mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0);
    if (mhz.empty()) return set_error(MI355_ERR_STATE, "clock probe: no stamps");
    std::nth_element(mhz.begin(), mhz.begin() + mhz.size() / 2, mhz.end());
    *shader_mhz = mhz[mhz.size() / 2];
    return MI355_OK;
}
