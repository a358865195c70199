// oracle/ref_harness/threads_synth.cpp -- synthetic frame source/sink for the reference server.
//
// TEST INFRASTRUCTURE ONLY.  Implements the reference's diff::threads::ThreadsCore interface
// (server/include/threads.hpp:37-45, included from the reference tree at build time -- the header
// is not copied here) without OpenCV, V4L2, pipes or sockets, so that the reference's own
// server/src/server.cpp main loop can be compiled *unmodified* and driven with seeded frames:
//
//   oracle/_ref/server_cpu : server.cpp with the CPU branch selected (server.cpp:78-135), used to
//                            pin the oracle's gray-avg -> histogram -> two-max -> binarize chain.
//   oracle/_ref/server_hip : server.cpp GPU branch linked against the CUDACore drop-in
//                            (cudavideostream_amd/compat) -- the integration check of the boundary.
//
// Frames come from the file named by $REF_IN and results go to $REF_OUT:
//   REF_IN : int32 width, int32 height, int32 nframes, base frame (3wh bytes), nframes frames
//   REF_OUT (CPU build): per frame the 3wh processed bytes handed to writeShow()
//   REF_OUT (HIP build): base frame, then per frame {u32 n, i32 xs[n], u8 diff[n]} -- the wire
//                        format of server/src/threads.cpp:223-233 -- and, if $REF_VIS is set, the
//                        visualisation frame of every iteration appended to that file.
// The process exits with status 0 from readCap() once the input is exhausted (the reference
// loop is `while (1)`).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "server/include/threads.hpp"
#ifdef SYNTH_HIP
#include "server/include/kernels.cuh"
#endif

using namespace diff::threads;

namespace {
struct synth_ctx {
    FILE *fin = nullptr, *fout = nullptr, *fvis = nullptr;
    int w = 0, h = 0, nframes = 0, served = 0, pass = 0;
    size_t total = 0;
    uint8_t *base = nullptr, *frame = nullptr, *vis = nullptr, *spare = nullptr;
    int *xs = nullptr;
    unsigned int hpos = 0;
};
synth_ctx *CTX(void *p) { return static_cast<synth_ctx *>(p); }

void die(const char *msg) {
    fprintf(stderr, "threads_synth: %s\n", msg);
    exit(2);
}
} // namespace

ThreadsCore::ThreadsCore() {
    synth_ctx *c = new synth_ctx;
    this->pctx = c;
    const char *in = getenv("REF_IN"), *out = getenv("REF_OUT"), *vis = getenv("REF_VIS");
    if (!in || !out) die("REF_IN / REF_OUT not set");
    c->fin = fopen(in, "rb");
    c->fout = fopen(out, "wb");
    if (!c->fin || !c->fout) die("cannot open REF_IN / REF_OUT");
    if (vis) c->fvis = fopen(vis, "wb");
    int32_t hdr[3];
    if (fread(hdr, sizeof hdr, 1, c->fin) != 1) die("short header");
    c->w = hdr[0]; c->h = hdr[1]; c->nframes = hdr[2];
    c->total = (size_t)3 * c->w * c->h;
    this->frameSz = diff::utils::matsz(c->h, c->w);
    // No glyph atlas (size 0x0): the reference's 1 Hz statistics string (server.cpp:151-171) may or
    // may not appear during a short run, and without glyphs the overlay is a no-op either way, so the
    // outputs stay deterministic.  The overlay itself is covered by tests/test_filters_gpu.py.
    this->charSz = diff::utils::matsz(0, 0);
    this->charsPx = new uint8_t[16];
    memset(this->charsPx, 0, 16);
#ifdef SYNTH_HIP
    diff::cuda::CUDACore::alloc_arrays(&c->frame, &c->vis, &c->spare, &c->xs, c->h, c->w);
#else
    c->frame = new uint8_t[c->total + 32];
    c->vis = new uint8_t[c->total + 32];
    c->xs = new int[c->total + 8];
#endif
    c->base = new uint8_t[c->total];
    if (fread(c->base, 1, c->total, c->fin) != c->total) die("short base frame");
    this->pbase = c->base;
    this->pshowready = c->vis;
#ifdef SYNTH_HIP
    fwrite(c->base, 1, c->total, c->fout); // threads.cpp:220 sends the base frame first
#endif
}

diff::utils::matsz ThreadsCore::getFrameSize() { return this->frameSz; }
diff::utils::matsz ThreadsCore::getCharSize() { return this->charSz; }
uint8_t *ThreadsCore::getCharsPx() { return this->charsPx; }
uint8_t *ThreadsCore::getBaseFrameData() { return CTX(pctx)->base; }
uint8_t *ThreadsCore::getShowReadyNData() { return CTX(pctx)->vis; }

void ThreadsCore::readCap(struct preadymin &minready) {
    synth_ctx *c = CTX(pctx);
    if (c->served == c->nframes) {
        // $REF_REPEAT = k: serve the sequence k times (timing runs; only the first pass is written out)
        const char *rep = getenv("REF_REPEAT");
        if (rep && ++c->pass < atoi(rep)) {
            fseek(c->fin, (long)(3 * sizeof(int32_t) + c->total), SEEK_SET);
            c->served = 0;
        } else {
            fclose(c->fout);
            if (c->fvis) fclose(c->fvis);
            exit(0);
        }
    }
    if (fread(c->frame, 1, c->total, c->fin) != c->total) die("short frame");
    c->served++;
    minready.data = c->frame;
    minready.h_pos = &c->hpos;
    minready.h_xs = c->xs;
    minready.__ptr = nullptr;
}

void ThreadsCore::writeNoise() {
    synth_ctx *c = CTX(pctx);
    if (c->pass) return;
    if (c->fvis) fwrite(c->vis, 1, c->total, c->fvis);
}

void ThreadsCore::writeShow(struct preadymin &minready) {
    synth_ctx *c = CTX(pctx);
    if (c->pass) return;
#ifdef SYNTH_HIP
    unsigned int n = *minready.h_pos;                          // threads.cpp:224-233
    fwrite(&n, sizeof n, 1, c->fout);
    fwrite(minready.h_xs, sizeof(int), n, c->fout);
    fwrite(minready.data, 1, n, c->fout);
#else
    fwrite(minready.data, 1, c->total, c->fout);
#endif
}
