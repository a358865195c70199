// compat/src/cudacore.cpp -- diff::cuda::CUDACore over the C-ABI of libmi355diff.so.
//
// Host code stays C++ (as in the reference) and reaches the GPU only through include/mi355diff.h.
// Error behaviour follows the reference's CUDA_CHECK (server/src/kernels.cu:11-22): message on stderr,
// then exit with a non-zero status; nothing is thrown and nothing is returned.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../../include/mi355diff.h"
#include "../include/common.h"
#include "../include/kernels.cuh"

using namespace diff::cuda;
using namespace diff::utils;

#define MI355_CHECK(call)                                                                         \
    do {                                                                                          \
        const int rc_ = (call);                                                                   \
        if (rc_ != MI355_OK) {                                                                    \
            fprintf(stderr, "MI355_CHECK() error %d (%s) @ %s:%d [%s]\n", rc_, mi355_last_error(), \
                    __FILE__, __LINE__, __func__);                                                \
            exit(rc_ < 0 ? -rc_ : rc_);                                                           \
        }                                                                                         \
    } while (0)

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

CUDACore::CUDACore(uint8_t *charsPx, matsz &charsSz, float *k, int total, uint8_t *sampleMatData,
                   matsz &frameSz) {
    memset(reserved_, 0, sizeof reserved_);
    reserved_int_ = 0;
    core_ = nullptr;
    total_ = total;
    if (total != 3 * frameSz.area()) {
        fprintf(stderr, "CUDACore: total (%d) != 3 * %d * %d\n", total, frameSz.height, frameSz.width);
        exit(1);
    }
    mi355_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.width = frameSz.width;
    cfg.height = frameSz.height;
    cfg.threshold = LR_THRESHOLDS;
    cfg.max_batch = 1;
    cfg.device = 0;  // kernels.cu:385 uses device 0
#ifdef NOISE_FILTER
    cfg.noise_filter = 1;
#endif
#ifdef NOISE_VISUALIZER
    cfg.visualizer = NOISE_VISUALIZER;
#endif
    cfg.noise_filter = env_int("MI355_NOISE_FILTER", cfg.noise_filter);
    cfg.visualizer = env_int("MI355_VISUALIZER", cfg.visualizer);
    MI355_CHECK(mi355_create(&cfg, &core_));
    if (k) MI355_CHECK(mi355_set_conv_kernel(core_, k));                         // kernels.cu:394
    if (charsPx && charsSz.area() > 0)                                            // kernels.cu:379-382
        MI355_CHECK(mi355_set_glyphs(core_, charsPx, (int)(sizeof(CHARS_STR) - 1), charsSz.height,
                                     charsSz.width, CHARS_STR));
    if (sampleMatData) MI355_CHECK(mi355_set_state(core_, sampleMatData));        // kernels.cu:406
    // nothing is left to be made, loaded or first-used inside exec_core (the reference allocates everything here too,
    // kernels.cu:395-402)
    MI355_CHECK(mi355_prepare(core_, MI355_PREPARE_EXEC | MI355_PREPARE_GRAY_CHAIN));
}

void CUDACore::exec_core(uint8_t *frameData, uint8_t *showReadyNData, std::string &text,
                         unsigned int *h_pos, int *h_xs) {
    uint32_t pos = 0;
    MI355_CHECK(mi355_exec(core_, frameData, showReadyNData, text.empty() ? nullptr : text.c_str(), &pos,
                           reinterpret_cast<int32_t *>(h_xs)));
    *h_pos = pos;
}

void CUDACore::pipe_open(int depth) { MI355_CHECK(mi355_pipe_open(core_, depth)); }

void CUDACore::pipe_close() { MI355_CHECK(mi355_pipe_close(core_)); }

long long CUDACore::exec_submit(uint8_t *frameData, uint8_t *showReadyNData, std::string &text, int *h_xs) {
    int64_t ticket = -1;
    MI355_CHECK(mi355_pipe_submit(core_, frameData, showReadyNData, text.empty() ? nullptr : text.c_str(),
                                  reinterpret_cast<int32_t *>(h_xs), &ticket));
    return ticket;
}

void CUDACore::exec_wait(long long ticket, unsigned int *h_pos) {
    uint32_t pos = 0;
    MI355_CHECK(mi355_pipe_wait(core_, ticket, &pos));
    *h_pos = pos;
}

size_t CUDACore::chunkt_size() { return 32; }  // sizeof(long4), kernels.cu:27,527-529

void CUDACore::alloc_arrays(uint8_t **h_frame, uint8_t **n_frame, uint8_t **o_frame, int **h_xs, int r,
                            int c) {
    const size_t n = (size_t)3 * r * c, slack = 32;
    MI355_CHECK(mi355_host_alloc((void **)h_frame, n + slack));
    MI355_CHECK(mi355_host_alloc((void **)n_frame, n + slack));
    MI355_CHECK(mi355_host_alloc((void **)o_frame, n + slack));
    MI355_CHECK(mi355_host_alloc((void **)h_xs, n * sizeof(int) + slack));
}
