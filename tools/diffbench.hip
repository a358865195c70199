// tools/diffbench.hip -- C++ harness over the C-ABI (include/mi355diff.h): the same workload as
// bench.py (S1 `webcam` stream, B frames resident in HBM, stateful diff+threshold+pack) without
// Python/PyTorch in the process.  Used for the rocprofv3 --pmc passes (the counter-collection
// interposer crashes under PyTorch's own kernels on this image) and as a plain C++ example of the
// boundary.  The synthetic generator is a device-side restatement of cudavideostream_amd/synth.py
// (tests/test_tools_gpu.py checks the two produce identical bytes).
//
//   diffbench [--width W] [--height H] [--batch B] [--steps K] [--warmup W] [--seed S]
//             [--pairs] [--checksum T] [--cores C] [--digest]
//             [--opt ID=VALUE ...]      mi355_set_option on every core (1 pipeline, 2 split per cent, 3 dense per cent,
//                                       4 chain hint, 5 pack workgroups, 6 median band rows: include/mi355diff.h "Options")
//             [--regime s0|flip|static] [--apart]   pairs of the dense / static regimes; pairs that share no frame
//             [--skew-frames N] [--skew-xs N] [--skew-df N] [--print-ptrs]   the frames / the two output arrays displaced by N
//                                       bytes inside a larger allocation
//             [--reroll N]              pairs: after the run, N new pairs of output arrays, N new index arrays, N new value
//                                       arrays, N new cores, N new copies of the frames -- the kernels' times after each
//                                       (which buffer's placement decides the dense expansion's speed: profiles/README.md)
//             [--lib-alloc]             the two output arrays from mi355_alloc_outputs (a pair placed for the dense expansion)
//             [--place N]               pairs: the two output arrays inside ONE allocation, the value array at a sweep of
//                                       distances behind the index array and the pair at a sweep of displacements; then N
//                                       pairs of arrays from hipMalloc, from the HIP virtual-memory calls (hipMemCreate: one
//                                       physical handle per array) and from mi355_dev_alloc -- the expansion's time for each
//   diffbench --filters [--batch B] [--steps K]     the filter kernels and the BASELINE config 3 / 4 chains
//                                                   (same lines as tools/bench_filters.py, for the --pmc passes)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "../include/mi355diff.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)
#define MI_OK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: %s\n", #x, mi355_last_error()); exit(4); } } while (0)

__host__ __device__ inline uint32_t hash32(uint32_t x) {  // lowbias32, synth.py hash32
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// synth.py webcam_frame(t, width, height, seed): t = -1 is the base frame.
__global__ void k_webcam_frame(uint8_t *out, int t, int width, int height, uint32_t seed) {
    const uint32_t n = 3u * width * height;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t tt = (uint32_t)(t + 1) & 0xFFFFu;
    const uint32_t key = hash32(hash32(idx) + tt * 0x9E3779B9u + seed * 0x85EBCA6Bu);
    const int nz = (int)((key & 0xFFFFu) % 17u) - 8;
    const uint32_t pix = idx / 3u;
    const int c = (int)(idx - pix * 3u);
    const int y = (int)(pix / (uint32_t)width), x = (int)(pix % (uint32_t)width);
    const int tex = (int)((hash32(idx + 0x5BD1E995u) >> 8) % 7u);
    int val = 40 + (x * 150) / width + (y * 40) / height + 5 * c + tex;
    if (t >= 0) {
        const int rw = width / 4, rh = height / 4;
        const int speed = width / 240 > 1 ? width / 240 : 1;
        const int x0 = (t * speed) % (width - rw), y0 = height / 3;
        if (x >= x0 && x < x0 + rw && y >= y0 && y < y0 + rh) val = 120 + 10 * c + (x - x0) / 16;
    }
    val += nz;
    if (t >= 0 && ((key >> 16) & 0xFFFFu) < 786u) val = (int)(hash32(key ^ 0xABCDEFu) & 0xFFu);
    out[idx] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
}

// synth.py refrand_frame(n, seed): S0, bytes uniform on 0..254 (the generator of tests/algorithms_benchmarks.cu:4-10);
// flip != 0: the same frame with bit 7 of every byte flipped (P = N against the unflipped one)
__global__ void k_refrand_frame(uint8_t *out, uint32_t n, uint32_t seed, int flip) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t v = hash32(hash32(idx) + seed * 0x85EBCA6Bu) % 255u;
    out[idx] = (uint8_t)(flip ? v ^ 0x80u : v);
}

// --corun valu|mem|lds: a background kernel of single-wave workgroups on a stream of its own while the timed batches run
// (what does the pack kernel share with a co-runner: issue slots or the memory system?).  Experiment only.
__global__ __launch_bounds__(64) void k_corun_valu(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u, b = blockIdx.x + 17u, c = a ^ 0x9e3779b9u, d = b + a;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { a = (a | 0x80808080u) - (b & 0x7f7f7f7fu); b = (b ^ c) + d; c = (c + a) ^ (d >> 3); d = d * 5u + b; }
    }
    if ((a ^ b ^ c ^ d) == 0x12345u) out[0] = a;
}
__global__ __launch_bounds__(64) void k_corun_mem(const uint4 *buf, size_t nvec, uint32_t *out, int iters) {
    uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 2654435761u, acc = 0;
    for (int i = 0; i < iters; i++) {   // every group of 4 lanes reads one random 64-byte piece
        x = x * 1664525u + 1013904223u;
        const uint32_t piece = __builtin_amdgcn_readfirstlane(0) + ((x >> 2) | 0u);
        const size_t idx = (((size_t)(piece ^ (threadIdx.x >> 2) * 0x9E3779B1u)) % (nvec / 4)) * 4 + (threadIdx.x & 3);
        const uint4 v = buf[idx];
        acc += v.x ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}

// --place probes: what distinguishes output arrays on which the dense expansion is fast from those on which it is slow?
typedef uint32_t pv4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_wide(pv4 *buf, size_t nvec) {   // streaming 16-byte non-temporal stores
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const pv4 v = {(uint32_t)i, 1u, 2u, 3u};
        __builtin_nontemporal_store(v, buf + i);
    }
}
// the dense expansion's own shape: one wave per item, an item = a contiguous run of E entries: 16 bytes of indices and 4 of
// values per lane and step, non-temporal
__global__ __launch_bounds__(64) void k_probe_items(uint32_t *xs, uint8_t *df, uint32_t E, uint32_t items_per_frame) {
    const size_t item = (size_t)blockIdx.y * items_per_frame + blockIdx.x;
    uint32_t *x = xs + item * E;
    uint8_t *d = df + item * E;
    for (uint32_t e = threadIdx.x * 4u; e + 3u < E; e += 256u) {
        const pv4 v = {e, e + 1u, e + 2u, e + 3u};
        __builtin_nontemporal_store(v, reinterpret_cast<pv4 *>(x + e));
        __builtin_nontemporal_store(e, reinterpret_cast<uint32_t *>(d + e));
    }
}
// the same with the items dealt to the workgroups in a scattered order (item = w * step mod n): the waves that run at the same
// time then write all over the arrays instead of inside one sliding window
__global__ __launch_bounds__(64) void k_probe_items_perm(uint32_t *xs, uint8_t *df, uint32_t E, uint32_t nitems, uint32_t step) {
    const size_t w = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (w >= nitems) return;
    const size_t item = (w * step) % nitems;
    uint32_t *x = xs + item * E;
    uint8_t *d = df + item * E;
    for (uint32_t e = threadIdx.x * 4u; e + 3u < E; e += 256u) {
        const pv4 v = {e, e + 1u, e + 2u, e + 3u};
        __builtin_nontemporal_store(v, reinterpret_cast<pv4 *>(x + e));
        __builtin_nontemporal_store(e, reinterpret_cast<uint32_t *>(d + e));
    }
}
// only the value array / only the index array of the expansion's shape
__global__ __launch_bounds__(64) void k_probe_items_one(uint32_t *xs, uint8_t *df, uint32_t E, uint32_t items_per_frame, int which) {
    const size_t item = (size_t)blockIdx.y * items_per_frame + blockIdx.x;
    uint32_t *x = xs + item * E;
    uint8_t *d = df + item * E;
    for (uint32_t e = threadIdx.x * 4u; e + 3u < E; e += 256u) {
        const pv4 v = {e, e + 1u, e + 2u, e + 3u};
        if (which == 0) __builtin_nontemporal_store(v, reinterpret_cast<pv4 *>(x + e));
        else __builtin_nontemporal_store(e, reinterpret_cast<uint32_t *>(d + e));
    }
}
// one random 64-byte line per lane and step: address translation (TLB reach: page-fragment size) and row misses, no streaming
__global__ __launch_bounds__(256) void k_probe_rand(const pv4 *buf, size_t nlines, uint32_t *sink, int iters) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const size_t line = ((size_t)x * 2654435761ull >> 7) % nlines;
        acc += buf[line * 4].x;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char **argv) {
    int W = 1920, H = 1080, B = 256, K = 20, WU = 3, checksum_t = -2, ncores = 1;
    uint32_t seed = 21;
    bool pairs = false, filters = false, digest = false, apart = false, print_ptrs = false;
    size_t skew_xs = 0, skew_df = 0, skew_frames = 0;
    int reroll = 0, place = 0;
    bool lib_alloc = false;   // the two output arrays from mi355_alloc_outputs (a pair placed for the dense expansion)
    const char *corun = nullptr; int corun_blocks = 2048;
    std::vector<std::pair<int, int>> opts;
    const char *regime = nullptr;   // --regime s0|flip|static: pairs of the dense / static regimes (tools/bench_regimes.py's inputs)
    for (int i = 1; i < argc; i++) {
        auto next = [&](int &v) { if (i + 1 < argc) v = atoi(argv[++i]); };
        if (!strcmp(argv[i], "--width")) next(W);
        else if (!strcmp(argv[i], "--height")) next(H);
        else if (!strcmp(argv[i], "--batch")) next(B);
        else if (!strcmp(argv[i], "--steps")) next(K);
        else if (!strcmp(argv[i], "--warmup")) next(WU);
        else if (!strcmp(argv[i], "--seed")) { int s = 21; next(s); seed = (uint32_t)s; }
        else if (!strcmp(argv[i], "--checksum")) next(checksum_t);
        else if (!strcmp(argv[i], "--pairs")) pairs = true;
        else if (!strcmp(argv[i], "--apart")) { pairs = true; apart = true; }   // pairs (0,1), (2,3), ...: no frame is an operand twice (round-robin sharding)
        else if (!strcmp(argv[i], "--cores")) next(ncores);
        else if (!strcmp(argv[i], "--filters")) filters = true;
        else if (!strcmp(argv[i], "--digest")) digest = true;
        else if (!strcmp(argv[i], "--corun") && i + 1 < argc) corun = argv[++i];
        else if (!strcmp(argv[i], "--regime") && i + 1 < argc) { regime = argv[++i]; pairs = true; }
        else if (!strcmp(argv[i], "--corun-blocks")) next(corun_blocks);
        else if (!strcmp(argv[i], "--skew-xs") && i + 1 < argc) skew_xs = (size_t)atoll(argv[++i]) & ~(size_t)15;
        else if (!strcmp(argv[i], "--skew-df") && i + 1 < argc) skew_df = (size_t)atoll(argv[++i]) & ~(size_t)15;
        else if (!strcmp(argv[i], "--skew-frames") && i + 1 < argc) skew_frames = (size_t)atoll(argv[++i]) & ~(size_t)15;
        else if (!strcmp(argv[i], "--print-ptrs")) print_ptrs = true;
        else if (!strcmp(argv[i], "--reroll")) next(reroll);
        else if (!strcmp(argv[i], "--place")) next(place);
        else if (!strcmp(argv[i], "--lib-alloc")) lib_alloc = true;
        else if (!strcmp(argv[i], "--opt") && i + 1 < argc) { int id = 0, v = 0; if (sscanf(argv[++i], "%d=%d", &id, &v) == 2) opts.push_back({id, v}); }
    }
    auto apply_opts = [&](mi355_core *c) { for (auto &o : opts) MI_OK(mi355_set_option(c, o.first, o.second)); };
    const size_t n = (size_t)3 * W * H;
    if (checksum_t >= -1) {  // print a checksum of one generated frame (generator cross-check)
        uint8_t *d; HIP_OK(hipMalloc((void **)&d, n));
        hipLaunchKernelGGL(k_webcam_frame, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, checksum_t, W, H, seed);
        std::vector<uint8_t> h(n);
        HIP_OK(hipMemcpy(h.data(), d, n, hipMemcpyDeviceToHost));
        uint64_t s1 = 0, s2 = 0;
        for (size_t i = 0; i < n; i++) { s1 += h[i]; s2 += (uint64_t)h[i] * (uint64_t)(i % 65521 + 1); }
        printf("{\"frame\": %d, \"sum\": %llu, \"wsum\": %llu}\n", checksum_t, (unsigned long long)s1, (unsigned long long)s2);
        return 0;
    }

    mi355_config cfg{};
    cfg.width = W; cfg.height = H; cfg.threshold = 20; cfg.max_batch = B; cfg.device = -1;
    if (ncores > 1) {   // several independent streams, one core (own HIP stream) each, submitted round-robin
        std::vector<mi355_core *> cores(ncores);
        std::vector<uint8_t *> fr(ncores);
        std::vector<uint32_t *> off(ncores);
        std::vector<int32_t *> xs(ncores);
        std::vector<uint8_t *> df(ncores);
        const size_t cap = (size_t)B * n / 8 > (1u << 20) ? (size_t)B * n / 8 : (1u << 20);
        const dim3 g((unsigned)((n + 255) / 256)), b(256);
        std::vector<uint8_t> h_base(n);
        for (int c = 0; c < ncores; c++) {
            MI_OK(mi355_create(&cfg, &cores[c]));
            apply_opts(cores[c]);
            HIP_OK(hipMalloc((void **)&fr[c], n * (size_t)(B + 1)));
            for (int t = -1; t < B; t++)
                hipLaunchKernelGGL(k_webcam_frame, g, b, 0, 0, fr[c] + (size_t)(t + 1) * n, t, W, H, seed + c);
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipMemcpy(h_base.data(), fr[c], n, hipMemcpyDeviceToHost));
            MI_OK(mi355_set_state(cores[c], h_base.data()));
            HIP_OK(hipMalloc((void **)&off[c], sizeof(uint32_t) * (B + 1)));
            HIP_OK(hipMalloc((void **)&xs[c], sizeof(int32_t) * cap));
            HIP_OK(hipMalloc((void **)&df[c], cap));
        }
        auto round = [&]() {
            for (int c = 0; c < ncores; c++)
                MI_OK(mi355_diff_stream_batch(cores[c], fr[c] + n, n, B, off[c], xs[c], df[c], cap));
        };
        for (int i = 0; i < WU; i++) round();
        for (int c = 0; c < ncores; c++) MI_OK(mi355_synchronize(cores[c]));
        const auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < K; i++) round();
        for (int c = 0; c < ncores; c++) MI_OK(mi355_synchronize(cores[c]));
        const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        printf("{\"harness\": \"diffbench\", \"mode\": \"stream\", \"cores\": %d, \"batch\": %d, \"steps\": %d, "
               "\"frames_per_s\": %.1f, \"ms_per_round\": %.4f}\n",
               ncores, B, K, (double)ncores * B * K / sec, sec / K * 1e3);
        for (auto c : cores) mi355_destroy(c);
        return 0;
    }
    if (filters) {   // mi355_filter_batch per kernel, then the two chains; B frames resident, K repetitions each
        mi355_core *core = nullptr;
        MI_OK(mi355_create(&cfg, &core));
        apply_opts(core);
        uint8_t *fr, *out, *filt; uint32_t *off; int32_t *xs; uint8_t *df;
        const size_t cap = (size_t)B * n / 4;
        HIP_OK(hipMalloc((void **)&fr, n * (size_t)(B + 1)));
        HIP_OK(hipMalloc((void **)&out, n * (size_t)B));
        HIP_OK(hipMalloc((void **)&filt, n * (size_t)B));
        HIP_OK(hipMalloc((void **)&off, sizeof(uint32_t) * (B + 1)));
        HIP_OK(hipMalloc((void **)&xs, sizeof(int32_t) * cap));
        HIP_OK(hipMalloc((void **)&df, cap));
        const dim3 g((unsigned)((n + 255) / 256)), b(256);
        for (int t = 0; t <= B; t++) hipLaunchKernelGGL(k_webcam_frame, g, b, 0, 0, fr + (size_t)t * n, t - 1, W, H, seed);
        HIP_OK(hipDeviceSynchronize());
        std::vector<uint8_t> h_base(n);
        HIP_OK(hipMemcpy(h_base.data(), fr, n, hipMemcpyDeviceToHost));
        MI_OK(mi355_set_state(core, h_base.data()));
        float k9[9];   // a normalised 3x3 Gaussian, sigma 1.5 (symmetric like the server's)
        { double s = 0; for (int i = 0; i < 9; i++) { const int y = i / 3 - 1, x = i % 3 - 1; k9[i] = (float)exp(-(x * x + y * y) / 4.5); s += k9[i]; }
          for (float &v : k9) v = (float)(v / s); }
        MI_OK(mi355_set_conv_kernel(core, k9));
        const uint8_t *cur = fr + n, *prev = fr;
        struct Op { const char *name; int op; bool two; double alg; };
        const double N = (double)n;
        const Op ops[] = {{"gray_avg", MI355_OP_GRAY_AVG, false, 2 * N}, {"gray_weighted", MI355_OP_GRAY_WEIGHTED, false, 2 * N},
                          {"gray_weighted+binarize fused (config 3 visualiser)", MI355_OP_GRAY_WEIGHTED_BINARIZE, false, 3 * N},
                          {"heat_map", MI355_OP_HEAT_MAP, true, 3 * N}, {"red_dense", MI355_OP_RED_DENSE, true, 3 * N},
                          {"conv3x3", MI355_OP_CONV3X3, false, 2 * N}, {"median5x5", MI355_OP_MEDIAN5X5, false, 2 * N}};
        auto timed = [&](const char *kind, const char *name, double alg, auto fn) {
            for (int i = 0; i < 2; i++) fn();
            MI_OK(mi355_synchronize(core));
            const auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < K; i++) fn();
            MI_OK(mi355_synchronize(core));
            const double us = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count() * 1e6 / ((double)K * B);
            printf("{\"%s\": \"%s\", \"us_per_frame\": %.3f, \"algorithmic_bytes_per_frame\": %.0f, \"achieved_gbps\": %.1f, \"frac_of_8TBps\": %.4f, \"batch\": %d}\n",
                   kind, name, us, alg, alg / (us * 1e-6) / 1e9, alg / (us * 1e-6) / 1e9 / 8000.0, B);
        };
        for (const Op &o : ops)
            timed("filter", o.name, o.alg, [&]() { MI_OK(mi355_filter_batch(core, o.op, cur, o.two ? prev : nullptr, out, n, B)); });
        uint32_t h_tot = 0;
        timed("chain", "config 3: gray-weighted + binarize + diff/threshold/pack", 4 * N, [&]() {
            MI_OK(mi355_filter_batch(core, MI355_OP_GRAY_WEIGHTED_BINARIZE, cur, nullptr, out, n, B));
            MI_OK(mi355_diff_stream_batch(core, cur, n, B, off, xs, df, cap)); });
        timed("chain", "config 4: noise filter + diff/threshold/pack + red motion map", 5 * N, [&]() {
            MI_OK(mi355_filter_batch(core, MI355_OP_CONV3X3, cur, nullptr, filt, n, B));
            MI_OK(mi355_diff_stream_batch(core, filt, n, B, off, xs, df, cap));
            MI_OK(mi355_red_stream_batch(core, off, xs, B, out, n, 1)); });
        HIP_OK(hipMemcpy(&h_tot, off + B, sizeof h_tot, hipMemcpyDeviceToHost));
        printf("{\"note\": \"chains: algorithmic bytes above exclude the 5P of the packed stream\", \"last_chain_changed_bytes_per_frame\": %.1f}\n", (double)h_tot / B);
        mi355_destroy(core);
        return 0;
    }
    mi355_core *core = nullptr;
    MI_OK(mi355_create(&cfg, &core));
    apply_opts(core);

    uint8_t *d_frames = nullptr, *d_base = nullptr;
    const int nfr = apart ? 2 * B : B + 1;
    HIP_OK(hipMalloc((void **)&d_frames, n * (size_t)nfr + skew_frames));
    d_frames += skew_frames;
    HIP_OK(hipMalloc((void **)&d_base, n));
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    hipLaunchKernelGGL(k_webcam_frame, g, b, 0, 0, d_base, -1, W, H, seed);
    for (int t = 0; t < nfr; t++)
        hipLaunchKernelGGL(k_webcam_frame, g, b, 0, 0, d_frames + (size_t)t * n, t, W, H, seed);
    HIP_OK(hipDeviceSynchronize());
    std::vector<uint8_t> h_base(n);
    HIP_OK(hipMemcpy(h_base.data(), d_base, n, hipMemcpyDeviceToHost));
    MI_OK(mi355_set_state(core, h_base.data()));

    uint8_t *d_cur = d_frames + n, *d_prev = d_frames;   // pairs: consecutive frames of the stream
    if (regime) {   // cur = refrand(100 + B + t) (s0) / prev ^ 0x80 (flip) / prev (static), prev = refrand(100 + t)
        uint8_t *d_r;
        HIP_OK(hipMalloc((void **)&d_r, n * (size_t)(2 * B)));
        for (int t = 0; t < B; t++) {
            hipLaunchKernelGGL(k_refrand_frame, g, b, 0, 0, d_r + (size_t)t * n, (uint32_t)n, 100u + t, 0);
            const bool s0 = !strcmp(regime, "s0");
            hipLaunchKernelGGL(k_refrand_frame, g, b, 0, 0, d_r + (size_t)(B + t) * n, (uint32_t)n, s0 ? 100u + B + t : 100u + t,
                               !strcmp(regime, "flip") ? 1 : 0);
        }
        HIP_OK(hipDeviceSynchronize());
        d_prev = d_r; d_cur = d_r + (size_t)B * n;
    }
    const size_t cap = regime ? (size_t)B * n : ((size_t)B * n / 8 > (1u << 20) ? (size_t)B * n / 8 : (1u << 20));
    uint32_t *d_off; int32_t *d_xs; uint8_t *d_df;
    HIP_OK(hipMalloc((void **)&d_off, sizeof(uint32_t) * (B + 1)));
    // --skew-xs / --skew-df BYTES (multiples of 16): the output arrays displaced inside a larger allocation -- does the
    // expansion's time depend on WHERE its outputs lie (profiles/r05ae_*)?  --print-ptrs shows the addresses.
    if (lib_alloc) {
        void *a = nullptr, *b2 = nullptr; int draws = 0;
        const auto ta = std::chrono::high_resolution_clock::now();
        MI_OK(mi355_alloc_outputs(core, cap, &a, &b2, &draws));
        fprintf(stderr, "mi355_alloc_outputs: %d value array(s) drawn in %.1f ms\n", draws, std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - ta).count() * 1e3);
        d_xs = (int32_t *)a; d_df = (uint8_t *)b2;
    } else {
    HIP_OK(hipMalloc((void **)&d_xs, sizeof(int32_t) * cap + skew_xs));
    HIP_OK(hipMalloc((void **)&d_df, cap + skew_df));
    }
    d_xs = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(d_xs) + skew_xs);
    d_df += skew_df;
    if (print_ptrs) fprintf(stderr, "d_xs %p d_df %p d_off %p frames %p\n", (void *)d_xs, (void *)d_df, (void *)d_off, (void *)(pairs ? d_cur : d_frames));

    auto step = [&]() {
        if (pairs) MI_OK(mi355_diff_pairs_batch(core, d_cur, d_prev, apart ? 2 * n : n, B, d_off, d_xs, d_df, cap));
        else MI_OK(mi355_diff_stream_batch(core, d_frames, n, B, d_off, d_xs, d_df, cap));
    };
    for (int i = 0; i < WU; i++) step();
    MI_OK(mi355_synchronize(core));
    MI_OK(mi355_set_timing(core, getenv("DIFFBENCH_NO_TIMING") ? 0 : 1));   // per-kernel HIP events (5 per batch) off: what do they cost?
    MI_OK(mi355_reset_timing(core));
    hipStream_t cs = nullptr; uint32_t *d_sink = nullptr; uint4 *d_big = nullptr; const size_t big = (size_t)2 << 30;
    if (corun) {
        HIP_OK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        HIP_OK(hipMalloc((void **)&d_sink, 64));
        if (!strcmp(corun, "mem")) { HIP_OK(hipMalloc((void **)&d_big, big)); HIP_OK(hipMemset(d_big, 1, big)); }
        // long enough to cover the timed region (a few tens of ms)
        if (!strcmp(corun, "valu")) hipLaunchKernelGGL(k_corun_valu, dim3(corun_blocks), dim3(64), 0, cs, d_sink, 400000);
        else hipLaunchKernelGGL(k_corun_mem, dim3(corun_blocks), dim3(64), 0, cs, d_big, big / 16, d_sink, 20000);
    }
    const auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < K; i++) step();
    MI_OK(mi355_synchronize(core));
    const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    if (corun) {
        const bool still = hipStreamQuery(cs) == hipErrorNotReady;   // the co-runner must have covered the whole timed region
        HIP_OK(hipStreamSynchronize(cs));
        const double csec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        fprintf(stderr, "corun %s: %d blocks, covered the timed region: %s, ran %.1f ms ", corun, corun_blocks, still ? "yes" : "NO", csec * 1e3);
    }
    double ms_pack = 0, ms_total = 0; int launches = 0;
    MI_OK(mi355_get_timing(core, &ms_pack, &ms_total, &launches));
    std::vector<uint32_t> off(B + 1);
    HIP_OK(hipMemcpy(off.data(), d_off, sizeof(uint32_t) * (B + 1), hipMemcpyDeviceToHost));
    const double p = off[B];
    if (digest) {   // order-sensitive digest of everything the batch produced (A/B builds must agree)
        std::vector<int32_t> hx((size_t)off[B]); std::vector<uint8_t> hd((size_t)off[B]);
        HIP_OK(hipMemcpy(hx.data(), d_xs, sizeof(int32_t) * hx.size(), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hd.data(), d_df, hd.size(), hipMemcpyDeviceToHost));
        uint64_t h = 1469598103934665603ull;
        for (int t = 0; t <= B; t++) h = (h ^ off[t]) * 1099511628211ull;
        for (size_t i = 0; i < hx.size(); i++) h = (h ^ ((uint64_t)(uint32_t)hx[i] << 8 | hd[i])) * 1099511628211ull;
        fprintf(stderr, "digest %016llx\n", (unsigned long long)h);
    }
    const double pack_ms = ms_pack / (launches ? launches : 1);
    const double alg = 2.0 * n * B + 5.0 * p;
    double k_pack = 0, k_scan = 0, k_exp = 0; int kl = 0;
    MI_OK(mi355_get_kernel_timing(core, &k_pack, &k_scan, &k_exp, &kl));
    double mhz = 0;
    if (!getenv("DIFFBENCH_NO_CLOCK")) MI_OK(mi355_probe_clock(core, 50, &mhz));   // right behind the timed region: the chip is warm
    double hbm = 0;
    if (getenv("DIFFBENCH_HBM_PROBE")) MI_OK(mi355_probe_hbm_read(core, 2048, &hbm));
    printf("{\"harness\": \"diffbench\", \"mode\": \"%s\", \"width\": %d, \"height\": %d, \"batch\": %d, \"steps\": %d, "
           "\"frames_per_s\": %.1f, \"ms_per_step\": %.4f, \"frac\": %.4f, \"kernel_ms\": %.4f, \"all_kernels_ms\": %.4f, \"kernels_us\": [%.1f, %.1f, %.1f], "
           "\"changed_bytes_per_frame\": %.1f, \"achieved_gbps\": %.1f, \"shader_mhz\": %.0f, \"hbm_read_probe_gbps\": %.0f, \"workspace_bytes\": %zu}\n",
           pairs ? "pairs" : "stream", W, H, B, K, (double)B * K / sec, sec / K * 1e3, alg / (sec / K) / 8e12, pack_ms, ms_total / (launches ? launches : 1),
           k_pack / (kl ? kl : 1) * 1e3, k_scan / (kl ? kl : 1) * 1e3, k_exp / (kl ? kl : 1) * 1e3,
           p / B, alg / (pack_ms * 1e-3) / 1e9, mhz, hbm, mi355_workspace_bytes(core));
    if (reroll > 0 && pairs) {
        // Which buffers' placement decides the dense expansion's speed (profiles/README.md, "two speeds")?  In ONE process:
        // new output arrays (the old ones stay allocated), then a new core = new logs (created before the old one is
        // destroyed), then new copies of the input frames; the three kernels' times after every re-draw.
        auto measure = [&](const char *what, int i) {
            for (int w = 0; w < WU; w++) MI_OK(mi355_diff_pairs_batch(core, d_cur, d_prev, apart ? 2 * n : n, B, d_off, d_xs, d_df, cap));
            MI_OK(mi355_synchronize(core));
            MI_OK(mi355_set_timing(core, 1));
            MI_OK(mi355_reset_timing(core));
            for (int k = 0; k < K; k++) MI_OK(mi355_diff_pairs_batch(core, d_cur, d_prev, apart ? 2 * n : n, B, d_off, d_xs, d_df, cap));
            MI_OK(mi355_synchronize(core));
            double a = 0, b = 0, c = 0; int l = 0;
            MI_OK(mi355_get_kernel_timing(core, &a, &b, &c, &l));
            printf("reroll %s %d: kernels_us [%.1f, %.1f, %.1f]  xs %p df %p\n", what, i, a / l * 1e3, b / l * 1e3, c / l * 1e3, (void *)d_xs, (void *)d_df);
        };
        for (int i = 1; i <= reroll; i++) {
            HIP_OK(hipMalloc((void **)&d_xs, sizeof(int32_t) * cap));
            HIP_OK(hipMalloc((void **)&d_df, cap));
            measure("outputs", i);
        }
        for (int i = 1; i <= reroll; i++) {   // the index array alone, then the value array alone
            HIP_OK(hipMalloc((void **)&d_xs, sizeof(int32_t) * cap));
            measure("xs-only", i);
        }
        for (int i = 1; i <= reroll; i++) {
            HIP_OK(hipMalloc((void **)&d_df, cap));
            measure("df-only", i);
        }
        for (int i = 1; i <= reroll; i++) {
            mi355_core *fresh = nullptr;
            MI_OK(mi355_create(&cfg, &fresh));
            mi355_destroy(core);
            core = fresh;
            apply_opts(core);
            measure("core", i);
        }
        for (int i = 1; i <= reroll; i++) {
            const size_t bytes = (size_t)((regime || apart) ? 2 * B : B + 1) * n;
            uint8_t *copy = nullptr;
            HIP_OK(hipMalloc((void **)&copy, bytes));
            HIP_OK(hipMemcpy(copy, d_prev, bytes, hipMemcpyDeviceToDevice));
            d_cur = copy + (d_cur - d_prev);
            d_prev = copy;
            measure("frames", i);
        }
    }
    if (place > 0 && pairs) {
        // Where do the two output arrays have to lie for the dense expansion to run at its fast speed (profiles/README.md, r06)?
        auto measure = [&](const char *what, long long a, long long b2) {
            for (int w = 0; w < WU; w++) MI_OK(mi355_diff_pairs_batch(core, d_cur, d_prev, apart ? 2 * n : n, B, d_off, d_xs, d_df, cap));
            MI_OK(mi355_synchronize(core));
            MI_OK(mi355_set_timing(core, 1));
            MI_OK(mi355_reset_timing(core));
            for (int k = 0; k < K; k++) MI_OK(mi355_diff_pairs_batch(core, d_cur, d_prev, apart ? 2 * n : n, B, d_off, d_xs, d_df, cap));
            MI_OK(mi355_synchronize(core));
            double ta = 0, tb = 0, tc = 0; int l = 0;
            MI_OK(mi355_get_kernel_timing(core, &ta, &tb, &tc, &l));
            printf("place %s %lld %lld: expand_us %.1f pack_us %.1f  xs %p df %p\n", what, a, b2, tc / l * 1e3, ta / l * 1e3, (void *)d_xs, (void *)d_df);
            fflush(stdout);
        };
        const size_t xs_bytes = sizeof(int32_t) * cap, df_bytes = cap, MiB2 = (size_t)2 << 20;
        const size_t xs_span = (xs_bytes + MiB2 - 1) / MiB2 * MiB2;
        uint8_t *block = nullptr;
        HIP_OK(hipMalloc((void **)&block, xs_span + df_bytes + 64 * MiB2));
        printf("place block %p (%zu bytes): xs %zu bytes, df %zu bytes\n", (void *)block, xs_span + df_bytes + 64 * MiB2, xs_bytes, df_bytes);
        const long long rel[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20};
        const bool sweeps = !getenv("DIFFBENCH_PLACE_NO_SWEEPS");
        for (int rep = 0; rep < 2 && sweeps; rep++)
            for (long long r : rel) {
                d_xs = (int32_t *)block; d_df = block + xs_span + r;
                measure("rel", r, rep);
            }
        const long long dis[] = {0, 4096, 65536, 1 << 20, 2 << 20, 5 << 20, 16 << 20};
        for (long long d : dis) {
            if (!sweeps) break;
            d_xs = (int32_t *)(block + d); d_df = block + xs_span + (32 << 20) + d;
            measure("both", d, 0);
        }
        // separate allocations, kept allocated: the expansion on each pair, then three probes on the same memory
        hipEvent_t pe0, pe1; HIP_OK(hipEventCreate(&pe0)); HIP_OK(hipEventCreate(&pe1));
        uint32_t *psink; HIP_OK(hipMalloc((void **)&psink, 64));
        auto timed_us = [&](auto fn) {
            float best = 1e30f;
            for (int r = 0; r < 4; r++) {
                HIP_OK(hipEventRecord(pe0, 0)); fn(); HIP_OK(hipEventRecord(pe1, 0)); HIP_OK(hipEventSynchronize(pe1));
                float ms = 0; HIP_OK(hipEventElapsedTime(&ms, pe0, pe1));
                if (r > 0 && ms < best) best = ms;
            }
            return best * 1e3;
        };
        const uint32_t items = (uint32_t)((n / 1024 + 15) / 16), E = (uint32_t)(cap / ((size_t)items * B)) & ~3u;
        auto probes = [&](const char *what, int i) {
            MI_OK(mi355_synchronize(core));
            const double wide_xs = timed_us([&] { hipLaunchKernelGGL(k_probe_wide, dim3(2048), dim3(256), 0, 0, (pv4 *)d_xs, xs_bytes / 16); });
            const double wide_df = timed_us([&] { hipLaunchKernelGGL(k_probe_wide, dim3(2048), dim3(256), 0, 0, (pv4 *)d_df, df_bytes / 16); });
            const double it = timed_us([&] { hipLaunchKernelGGL(k_probe_items, dim3(items, B), dim3(64), 0, 0, (uint32_t *)d_xs, d_df, E, items); });
            const double itp = timed_us([&] { hipLaunchKernelGGL(k_probe_items_perm, dim3(items, B), dim3(64), 0, 0, (uint32_t *)d_xs, d_df, E, items * B, 4099u); });
            const double only_x = timed_us([&] { hipLaunchKernelGGL(k_probe_items_one, dim3(items, B), dim3(64), 0, 0, (uint32_t *)d_xs, d_df, E, items, 0); });
            const double only_d = timed_us([&] { hipLaunchKernelGGL(k_probe_items_one, dim3(items, B), dim3(64), 0, 0, (uint32_t *)d_xs, d_df, E, items, 1); });
            printf("probe2 %s %d: items_perm %.1f us  items_xs_only %.1f us  items_df_only %.1f us\n", what, i, itp, only_x, only_d);
            const double rx = timed_us([&] { hipLaunchKernelGGL(k_probe_rand, dim3(2048), dim3(256), 0, 0, (const pv4 *)d_xs, xs_bytes / 64, psink, 64); });
            const double rd = timed_us([&] { hipLaunchKernelGGL(k_probe_rand, dim3(2048), dim3(256), 0, 0, (const pv4 *)d_df, df_bytes / 64, psink, 64); });
            printf("probe %s %d: wide_xs %.1f us (%.0f GB/s) wide_df %.1f us (%.0f GB/s) items %.1f us (%.0f GB/s) rand_xs %.1f us rand_df %.1f us\n", what, i,
                   wide_xs, xs_bytes / wide_xs / 1e3, wide_df, df_bytes / wide_df / 1e3, it, 5.0 * E * items * B / it / 1e3, rx, rd);
            fflush(stdout);
        };
        const int burn_gb = getenv("DIFFBENCH_BURN_GB") ? atoi(getenv("DIFFBENCH_BURN_GB")) : 0;
        for (int g2 = 0; g2 < burn_gb; g2++) { void *b4 = nullptr; HIP_OK(hipMalloc(&b4, (size_t)1 << 30)); }
        if (burn_gb) printf("place burned %d GiB\n", burn_gb);
        std::vector<std::pair<int32_t *, uint8_t *>> kept;
        for (int i = 0; i < place; i++) {
            HIP_OK(hipMalloc((void **)&d_xs, xs_bytes));
            HIP_OK(hipMalloc((void **)&d_df, df_bytes));
            kept.push_back({d_xs, d_df});
            measure("hipMalloc", i, 0);
            probes("hipMalloc", i);
        }
        // the same arrays again in reverse order: is the speed a property of the MEMORY (it stays with the array)?
        for (int i = place - 1; i >= 0; i -= 3) { d_xs = kept[i].first; d_df = kept[i].second; measure("again", i, 0); }
        // mixed: index array of the last pair (the latest draw) with the value array of the first, and the other way round
        if (place > 1) {
            d_xs = kept[place - 1].first; d_df = kept[0].second; measure("mixed_xs_last_df_first", 0, 0);
            d_xs = kept[0].first; d_df = kept[place - 1].second; measure("mixed_xs_first_df_last", 0, 0);
        }
        // free everything and draw again: does a fresh draw get the same memory back?
        for (auto &kv : kept) { HIP_OK(hipFree(kv.first)); HIP_OK(hipFree(kv.second)); }
        for (int i = 0; i < 3; i++) {
            HIP_OK(hipMalloc((void **)&d_xs, xs_bytes));
            HIP_OK(hipMalloc((void **)&d_df, df_bytes));
            measure("after_free", i, 0);
        }
        // the value array put together from separately made physical pieces (HIP virtual-memory calls), mapped one after the
        // other or in a shuffled order: is scattered memory the fast kind?
        auto vmm_pieces = [&](size_t bytes, size_t piece, bool shuffle) -> void * {
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            int dev = 0; HIP_OK(hipGetDevice(&dev));
            prop.location.id = dev;
            const size_t np = (bytes + piece - 1) / piece;
            void *va = nullptr;
            HIP_OK(hipMemAddressReserve(&va, np * piece, (size_t)2 << 20, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> hs(np);
            for (size_t k = 0; k < np; k++) HIP_OK(hipMemCreate(&hs[k], piece, &prop, 0));
            std::vector<size_t> order(np);
            for (size_t k = 0; k < np; k++) order[k] = k;
            if (shuffle) { uint32_t r = 12345u; for (size_t k = np - 1; k > 0; k--) { r = r * 1664525u + 1013904223u; std::swap(order[k], order[(r >> 8) % (k + 1)]); } }
            for (size_t k = 0; k < np; k++) HIP_OK(hipMemMap((char *)va + k * piece, piece, 0, hs[order[k]], 0));
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            HIP_OK(hipMemSetAccess(va, np * piece, &acc, 1));
            return va;
        };
        HIP_OK(hipMalloc((void **)&d_xs, xs_bytes));
        const size_t pieces[] = {(size_t)2 << 20, (size_t)256 << 10, (size_t)64 << 10};
        for (size_t pc : pieces)
            for (int sh = 0; sh < 2; sh++)
                for (int i = 0; i < 2; i++) {
                    const auto t0p = std::chrono::high_resolution_clock::now();
                    d_df = (uint8_t *)vmm_pieces(df_bytes, pc, sh != 0);
                    const double ms_make = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0p).count() * 1e3;
                    char name[64]; snprintf(name, sizeof name, "vmm_%zuK_%s", pc >> 10, sh ? "shuffled" : "inorder");
                    printf("place %s made in %.1f ms\n", name, ms_make);
                    measure(name, i, 0);
                    probes(name, i);
                }
        for (int i = 0; i < 3; i++) {   // physically contiguous allocations
            void *a = nullptr, *b3 = nullptr;
            if (hipExtMallocWithFlags(&a, xs_bytes, hipDeviceMallocContiguous) != hipSuccess || hipExtMallocWithFlags(&b3, df_bytes, hipDeviceMallocContiguous) != hipSuccess) {
                printf("place contiguous: refused (%s)\n", hipGetErrorString(hipGetLastError())); break;
            }
            d_xs = (int32_t *)a; d_df = (uint8_t *)b3;
            measure("contiguous", i, 0);
            probes("contiguous", i);
        }
    }
    mi355_destroy(core);
    return 0;
}
