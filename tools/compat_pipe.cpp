// tools/compat_pipe.cpp -- the C++ drop-in class, blocking and pipelined, against each other.
//
// Plain C++ (g++) against cudavideostream_amd/compat: one diff::cuda::CUDACore runs a sequence of frames
// through exec_core (the reference's call, server/src/server.cpp:139), a second one through
// exec_submit / exec_wait with several frames in flight; h_pos, h_xs and the diff bytes of every frame
// must be identical.  Exit status 0 = identical.  Also prints what the FIRST exec_core of a fresh core takes against the
// median of the later ones (first_frame_ms / steady_frame_ms): nothing is allocated inside an entry point.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../cudavideostream_amd/compat/include/kernels.cuh"

using diff::cuda::CUDACore;
using diff::utils::matsz;

static uint32_t rs = 99;
static uint32_t rnd() { return rs = rs * 1664525u + 1013904223u; }

int main(int argc, char **argv) {
    int w = 640, h = 360, T = 12, depth = 3;
    if (argc > 2) { w = atoi(argv[1]); h = atoi(argv[2]); }
    if (argc > 3) T = atoi(argv[3]);
    const size_t n = (size_t)3 * w * h;
    std::vector<uint8_t> base(n);
    for (size_t i = 0; i < n; i++) base[i] = (uint8_t)(60 + (i * 11) % 120);
    std::vector<std::vector<uint8_t>> frames(T, std::vector<uint8_t>(n));
    for (int t = 0; t < T; t++)
        for (size_t i = 0; i < n; i++) {
            const uint32_t r = rnd();
            frames[t][i] = (r & 0xff) < 6 ? (uint8_t)(r >> 8) : (uint8_t)(base[i] + (r >> 24) % 7);
        }
    matsz chars(0, 0), fsz(h, w);
    uint8_t no_glyphs[16] = {0};
    float k[9] = {0};
    CUDACore blocking(no_glyphs, chars, k, (int)n, base.data(), fsz);      // server.cpp:53
    CUDACore piped(no_glyphs, chars, k, (int)n, base.data(), fsz);
    std::string text;

    // reference results, one frame at a time
    uint8_t *f, *nf, *of; int *xs;
    CUDACore::alloc_arrays(&f, &nf, &of, &xs, h, w);                        // threads.cpp:95
    std::vector<unsigned int> want_pos(T);
    std::vector<std::vector<int>> want_xs(T);
    std::vector<std::vector<uint8_t>> want_df(T);
    std::vector<double> ms(T);
    for (int t = 0; t < T; t++) {
        memcpy(f, frames[t].data(), n);
        unsigned int pos = 0;
        const auto t0 = std::chrono::steady_clock::now();
        blocking.exec_core(f, nf, text, &pos, xs);
        ms[t] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        want_pos[t] = pos;
        want_xs[t].assign(xs, xs + pos);
        want_df[t].assign(f, f + pos);
    }
    // the same frames with `depth` in flight, one pinned buffer set per slot
    struct Slot { uint8_t *f, *nf, *of; int *xs; long long ticket; int frame; };
    std::vector<Slot> ring(depth);
    for (auto &s : ring) { CUDACore::alloc_arrays(&s.f, &s.nf, &s.of, &s.xs, h, w); s.frame = -1; }
    piped.pipe_open(depth);
    int bad = 0;
    auto finish = [&](Slot &s) {
        unsigned int pos = 0;
        piped.exec_wait(s.ticket, &pos);
        const int t = s.frame;
        if (pos != want_pos[t] || memcmp(s.xs, want_xs[t].data(), pos * sizeof(int)) != 0 ||
            memcmp(s.f, want_df[t].data(), pos) != 0) {
            fprintf(stderr, "frame %d differs (h_pos %u vs %u)\n", t, pos, want_pos[t]);
            bad++;
        }
        s.frame = -1;
    };
    for (int t = 0; t < T; t++) {
        Slot &s = ring[t % depth];
        if (s.frame >= 0) finish(s);
        memcpy(s.f, frames[t].data(), n);
        s.frame = t;
        s.ticket = piped.exec_submit(s.f, s.nf, text, s.xs);
    }
    for (int t = T - depth < 0 ? 0 : T - depth; t < T; t++)
        if (ring[t % depth].frame == t) finish(ring[t % depth]);
    piped.pipe_close();
    unsigned long total = 0;
    for (int t = 0; t < T; t++) total += want_pos[t];
    const double first_ms = ms[0];
    std::vector<double> later(ms.begin() + (T > 1 ? 1 : 0), ms.end());
    std::sort(later.begin(), later.end());
    printf("{\"compat_pipe\": \"%s\", \"frames\": %d, \"depth\": %d, \"changed_bytes\": %lu, \"first_frame_ms\": %.3f, "
           "\"steady_frame_ms\": %.3f}\n", bad ? "MISMATCH" : "ok", T, depth, total, first_ms, later[later.size() / 2]);
    return bad ? 1 : 0;
}
