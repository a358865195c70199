/*
 * oracle/cpu_ref.h -- CPU restatement of the reference's frame-differencing + filter path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The shipped path (cudavideostream_amd/csrc, libmi355diff.so) never links or
 * calls it and has no CPU fallback.
 *
 * Every function cites the reference text it restates (paths relative to the reference repo
 * MatteoBattilana/CUDAVideoStream).  Pinning status is recorded per function in DESIGN.md
 * ("Oracle") and in tests/test_oracle_*.py.
 */
#ifndef ORACLE_CPU_REF_H_
#define ORACLE_CPU_REF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a-1: diff + threshold + negative feedback + pack ------------------------------------
 * tests/cuda_streaming/test.cu:560-576 (== server/src/server.cpp:82-94, commented).
 * state[] is `previous` on entry and the next `previous` on return (pvs in the reference).
 * xs/diff need capacity n.  Returns pos. */
uint32_t ora_diff_pack(const uint8_t *cur, uint8_t *state, size_t n, int thr,
                       int32_t *xs, uint8_t *diff);

/* Same loop, written the way the reference writes it: the diff bytes overwrite the head of the
 * frame buffer in place (test.cu:567 `pframe->data[h_pos] = df`).  frame[] is in/out. */
uint32_t ora_diff_pack_inplace(uint8_t *frame, uint8_t *state, size_t n, int thr, int32_t *xs);

/* Row-band-parallel form of ora_diff_pack over `nthreads` host threads (identical output:
 * bands are concatenated in order).  Used only as the all-cores CPU baseline in bench.py. */
uint32_t ora_diff_pack_mt(const uint8_t *cur, uint8_t *state, size_t n, int thr,
                          int32_t *xs, uint8_t *diff, int nthreads);

/* Stream of nframes frames through ora_diff_pack; offsets[nframes+1] = exclusive scan of the
 * per-frame counts; xs/diff are the per-frame outputs concatenated (capacity cap entries).
 * Returns 0, or -1 if cap is too small. */
int ora_diff_stream(const uint8_t *frames, int nframes, uint8_t *state, size_t n, int thr,
                    uint32_t *offsets, int32_t *xs, uint8_t *diff, size_t cap);
/* the same over all host cores: a row band per thread for all frames of the batch; identical output (bench.py's
 * cpu_baseline.all_cores).  0 ok, -1 cap too small, -2 out of memory / threads. */
int ora_diff_stream_mt(const uint8_t *frames, int nframes, uint8_t *state, size_t n, int thr,
                       uint32_t *offsets, int32_t *xs, uint8_t *diff, size_t cap, int nthreads);

/* client/opencv.cpp:64-66: frame[xs[i]] += diff[i] (uint8 wrap). */
void ora_client_apply(uint8_t *frame, const int32_t *xs, const uint8_t *diff, uint32_t n);

/* ---- a-5: integer diff benchmark ---------------------------------------------------------
 * tests/algorithms_benchmarks.cu:4-10 generateImage (glibc rand() % 255, here seeded),
 * :24-30 kernel1 (diff = cur - prev, no threshold), :12-22 checkDifference (indexing quirk
 * i*h+a with h=1920,w=1080 kept). */
void ora_generate_image(int32_t *image, int h, int w, unsigned seed);
void ora_int_diff(const int32_t *cur, const int32_t *prev, int32_t *diff, size_t n);
int ora_check_difference(const int32_t *f1, const int32_t *f2, const int32_t *d, int h, int w);

/* ---- a-6: noise filter -------------------------------------------------------------------
 * server/src/server.cpp:20-36 computeGaussianKernel (float sum, double intermediates). */
void ora_gaussian_kernel(float *k, int K, float sigma);
/* server/src/kernels.cu:97-136 convolution_kernel semantics, K=3: zero halo, float accumulator,
 * taps i-major/j-minor, one multiply then one add per tap (no FMA contraction), float->u8
 * truncation.  Hard-coded 1920x1080 and the halo channel-2 bug are not reproduced. */
void ora_conv3x3(const uint8_t *in, uint8_t *out, int w, int h, const float *k);
/* tests/noise_filter_benchmark/v2.cu:36-80 (K x K, K = 1..9, even K included: taps at -K/2 .. K-1-K/2) and
 * :116-124 computeMeanKernel; the Gaussian of the study is ora_gaussian_kernel (v2.cu:139-160 == server.cpp:20-36). */
void ora_conv_kxk(const uint8_t *in, uint8_t *out, int w, int h, const float *k, int K);
void ora_mean_kernel(float *k, int K);
/* Secondary oracle, tests/noise_filter_benchmark/cpu.cu:72-98 (int accumulator that truncates
 * after every tap; int images). */
/* tests/noise_filter_benchmark/v3.cu:32-90: 5x5 median per channel, zeros outside the image. */
void ora_median5x5(const uint8_t *in, uint8_t *out, int w, int h);
void ora_conv3x3_intacc(const int32_t *in, int32_t *out, int w, int h, const float *k);

/* ---- a-7: heat map -----------------------------------------------------------------------
 * tests/heat_map_benchmark/cpu.cu:19-27 getHeatPixel for d = 0..765 -> lut[d][0..2] = B,G,R;
 * :54-66 per-pixel loop (d = sum of per-channel |cur-prev|). */
void ora_heat_lut(uint8_t *lut /* 766*3 */);
void ora_heat_map(const uint8_t *cur, const uint8_t *prev, uint8_t *out, size_t npix);

/* ---- a-10: red motion map ----------------------------------------------------------------
 * tests/heat_map_red_benchmark/cpu.cu:38-55 (dense) and kernels.cu:273-281 (overlap, without the
 * h_pos/nMaxThreads truncation). */
void ora_red_dense(const uint8_t *cur, const uint8_t *prev, uint8_t *out, size_t npix, int thr);
void ora_red_overlap(uint8_t *img, const int32_t *xs, uint32_t n);

/* ---- a-8: grayscale ----------------------------------------------------------------------
 * avg: server/src/server.cpp:96-101; weighted: tests/grayscale-weighted/cpu.cu:40 (double,
 * left to right, truncation), value replicated into the 3 channels as the server does
 * (kernels.cu:88-90). */
void ora_gray_avg(const uint8_t *in, uint8_t *out, size_t npix);
void ora_gray_weighted(const uint8_t *in, uint8_t *out, size_t npix);
uint8_t ora_gray_weighted_px(uint8_t b, uint8_t g, uint8_t r);

/* ---- a-9: binarize chain -----------------------------------------------------------------
 * server/src/server.cpp:103-106 histogram over every 3rd byte, :108-127 two-max threshold with
 * clamp [50,200], :129-135 binarize. */
void ora_histogram(const uint8_t *gray3, size_t nbytes, int32_t *hist /*256*/);
int ora_two_max_threshold(const int32_t *hist /*256*/);
void ora_binarize(const uint8_t *in, uint8_t *out, size_t nbytes, int thr);
/* whole CPU branch server.cpp:96-135 in place on frame[]; returns the threshold used. */
int ora_server_cpu_branch(uint8_t *frame, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
