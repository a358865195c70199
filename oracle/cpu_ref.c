/*
 * oracle/cpu_ref.c -- CPU restatement of the reference's diff/threshold/pack + filter path.
 *
 * TEST INFRASTRUCTURE ONLY (see cpu_ref.h).  Plain C, no dependencies beyond libc/libm/pthread.
 * Build with -O2 -ffp-contract=off (oracle/Makefile): the floating-point rows depend on
 * "one multiply, then one add" evaluation order with no FMA contraction.
 *
 * Pinning (details in DESIGN.md):
 *   - ora_server_cpu_branch / gray_avg / histogram / two_max / binarize are checked against the
 *     reference's own server/src/server.cpp CPU branch compiled unmodified (oracle/_ref/server_cpu).
 *   - ora_diff_pack is pinned by the reference's identities: diff == cur-prev
 *     (tests/algorithms_benchmarks.cu:12-22), #(cur!=prev) == #(diff!=0)
 *     (tests/test_cuda/pixel_diff.cu:47-59) and the client reconstruction
 *     (client/opencv.cpp:64-66); the loop text itself only exists inside an OpenCV program.
 *   - heat / red / weighted gray / conv have no stored golden outputs upstream: restatement of
 *     the cited CPU text, fixtures generated here (tests/golden/).
 */
#define _GNU_SOURCE
#include "cpu_ref.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* a-1  tests/cuda_streaming/test.cu:560-576                                                    */

uint32_t ora_diff_pack(const uint8_t *cur, uint8_t *state, size_t n, int thr,
                       int32_t *xs, uint8_t *diff) {
    /* test.cu:561  Mat pvs = pframe->clone();  -> pvs starts as a copy of the current frame and
     * test.cu:571  pvs.data[i] -= df           -> un-flagged bytes fall back to previous[i].
     * test.cu:575  previous = pvs.  We update state[] in place: flagged -> cur, else unchanged. */
    uint32_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        int df = (int)cur[i] - (int)state[i];                 /* test.cu:565 */
        if (df < -thr || df > thr) {                          /* test.cu:566 */
            diff[pos] = (uint8_t)df;                          /* test.cu:567 */
            xs[pos] = (int32_t)i;                             /* test.cu:568 */
            pos++;                                            /* test.cu:569 */
            state[i] = cur[i];                                /* pvs keeps cur[i] */
        } else {
            state[i] = (uint8_t)(cur[i] - (uint8_t)df);       /* test.cu:571, == state[i] */
        }
    }
    return pos;
}

uint32_t ora_diff_pack_inplace(uint8_t *frame, uint8_t *state, size_t n, int thr, int32_t *xs) {
    uint32_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = frame[i];
        int df = (int)c - (int)state[i];
        if (df < -thr || df > thr) {
            frame[pos] = (uint8_t)df;   /* pos <= i, so the byte just read is the only one lost */
            xs[pos] = (int32_t)i;
            pos++;
            state[i] = c;
        }
    }
    return pos;
}

struct band_job {
    const uint8_t *cur;
    uint8_t *state;
    size_t lo, hi;
    int thr;
    int32_t *xs;   /* scratch, capacity hi-lo */
    uint8_t *diff; /* scratch */
    uint32_t count;
};

static void *band_worker(void *p) {
    struct band_job *j = (struct band_job *)p;
    uint32_t pos = 0;
    for (size_t i = j->lo; i < j->hi; i++) {
        int df = (int)j->cur[i] - (int)j->state[i];
        if (df < -j->thr || df > j->thr) {
            j->diff[pos] = (uint8_t)df;
            j->xs[pos] = (int32_t)i;
            pos++;
            j->state[i] = j->cur[i];
        }
    }
    j->count = pos;
    return NULL;
}

uint32_t ora_diff_pack_mt(const uint8_t *cur, uint8_t *state, size_t n, int thr,
                          int32_t *xs, uint8_t *diff, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    struct band_job jobs[256];
    pthread_t th[256];
    size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        size_t lo = (size_t)t * per, hi = lo + per;
        if (lo > n) lo = n;
        if (hi > n) hi = n;
        jobs[t].cur = cur; jobs[t].state = state; jobs[t].lo = lo; jobs[t].hi = hi;
        jobs[t].thr = thr; jobs[t].count = 0;
        /* each band packs into the tail-safe slot [lo, hi) of the output arrays, then bands are
         * compacted in order; pos <= i keeps the slots disjoint */
        jobs[t].xs = xs + lo; jobs[t].diff = diff + lo;
        pthread_create(&th[t], NULL, band_worker, &jobs[t]);
    }
    uint32_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].count && (size_t)total != jobs[t].lo) {
            memmove(xs + total, jobs[t].xs, (size_t)jobs[t].count * sizeof *xs);
            memmove(diff + total, jobs[t].diff, (size_t)jobs[t].count);
        }
        total += jobs[t].count;
    }
    return total;
}

int ora_diff_stream(const uint8_t *frames, int nframes, uint8_t *state, size_t n, int thr,
                    uint32_t *offsets, int32_t *xs, uint8_t *diff, size_t cap) {
    int32_t *txs = (int32_t *)malloc(n * sizeof *txs);
    uint8_t *tdf = (uint8_t *)malloc(n);
    if (!txs || !tdf) { free(txs); free(tdf); return -2; }
    size_t off = 0;
    offsets[0] = 0;
    for (int t = 0; t < nframes; t++) {
        uint32_t c = ora_diff_pack(frames + (size_t)t * n, state, n, thr, txs, tdf);
        if (off + c > cap) { free(txs); free(tdf); return -1; }
        memcpy(xs + off, txs, (size_t)c * sizeof *xs);
        memcpy(diff + off, tdf, c);
        off += c;
        offsets[t + 1] = (uint32_t)off;
    }
    free(txs); free(tdf);
    return 0;
}

/* The all-cores form of the baseline (bench.py, cpu_baseline.all_cores): the batch's frames through the SAME loop
 * (test.cu:560-576), a row band of the frame per thread for ALL frames of the batch -- a band's state belongs to its
 * thread, so the threads never meet between frames (ora_diff_pack_mt starts its threads anew for every frame) --
 * then the bands' pieces are put together in (frame, band) order, which is the ascending order of the whole frame:
 * output identical to ora_diff_stream (tests/test_oracle.py).  nthreads: 1..1024. */
struct stream_job {
    const uint8_t *frames; uint8_t *state;
    size_t n, lo, hi;
    int nframes, thr;
    int32_t *sxs; uint8_t *sdf;     /* the band's scratch: (hi - lo) * nframes entries at most */
    uint32_t *counts;               /* [nframes] entries of the band per frame */
    /* second phase */
    const size_t *dst;              /* [nframes] where the band's piece of frame f goes */
    int32_t *xs; uint8_t *diff;
    int failed;
};

static void *stream_pack_worker(void *p) {
    struct stream_job *j = (struct stream_job *)p;
    size_t pos = 0;
    for (int f = 0; f < j->nframes; f++) {
        const uint8_t *cur = j->frames + (size_t)f * j->n;
        const size_t first = pos;
        for (size_t i = j->lo; i < j->hi; i++) {
            int df = (int)cur[i] - (int)j->state[i];
            if (df < -j->thr || df > j->thr) {
                j->sdf[pos] = (uint8_t)df;
                j->sxs[pos] = (int32_t)i;
                pos++;
                j->state[i] = cur[i];
            }
        }
        j->counts[f] = (uint32_t)(pos - first);
    }
    return NULL;
}

static void *stream_place_worker(void *p) {
    struct stream_job *j = (struct stream_job *)p;
    size_t pos = 0;
    for (int f = 0; f < j->nframes; f++) {
        memcpy(j->xs + j->dst[f], j->sxs + pos, (size_t)j->counts[f] * sizeof *j->xs);
        memcpy(j->diff + j->dst[f], j->sdf + pos, j->counts[f]);
        pos += j->counts[f];
    }
    return NULL;
}

int ora_diff_stream_mt(const uint8_t *frames, int nframes, uint8_t *state, size_t n, int thr,
                       uint32_t *offsets, int32_t *xs, uint8_t *diff, size_t cap, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    if (nframes < 0) return -2;
    struct stream_job *jobs = (struct stream_job *)calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
    size_t *dst = (size_t *)calloc((size_t)nthreads * (size_t)(nframes ? nframes : 1), sizeof *dst);
    int rc = (jobs && th && dst) ? 0 : -2;
    const size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
    for (int t = 0; t < nthreads && rc == 0; t++) {
        struct stream_job *j = &jobs[t];
        j->frames = frames; j->state = state; j->n = n; j->nframes = nframes; j->thr = thr;
        j->lo = (size_t)t * per < n ? (size_t)t * per : n;
        j->hi = j->lo + per < n ? j->lo + per : n;
        const size_t room = (j->hi - j->lo) * (size_t)nframes + 1;   /* untouched pages cost nothing */
        j->sxs = (int32_t *)malloc(room * sizeof *j->sxs);
        j->sdf = (uint8_t *)malloc(room);
        j->counts = (uint32_t *)calloc((size_t)nframes + 1, sizeof *j->counts);
        j->dst = dst + (size_t)t * (size_t)(nframes ? nframes : 1);
        j->xs = xs; j->diff = diff;
        if (!j->sxs || !j->sdf || !j->counts) rc = -2;
    }
    if (rc == 0) {
        int started = 0;
        for (; started < nthreads; started++)
            if (pthread_create(&th[started], NULL, stream_pack_worker, &jobs[started]) != 0) { rc = -2; break; }
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    if (rc == 0) {   /* where every (frame, band) piece goes: frame-major, bands in order */
        size_t off = 0;
        offsets[0] = 0;
        for (int f = 0; f < nframes; f++) {
            for (int t = 0; t < nthreads; t++) {
                dst[(size_t)t * (size_t)nframes + (size_t)f] = off;
                off += jobs[t].counts[f];
            }
            offsets[f + 1] = (uint32_t)off;
        }
        if (off > cap) rc = -1;
    }
    if (rc == 0) {
        int started = 0;
        for (; started < nthreads; started++)
            if (pthread_create(&th[started], NULL, stream_place_worker, &jobs[started]) != 0) { rc = -2; break; }
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    if (jobs)
        for (int t = 0; t < nthreads; t++) { free(jobs[t].sxs); free(jobs[t].sdf); free(jobs[t].counts); }
    free(jobs); free(th); free(dst);
    return rc;
}

/* client/opencv.cpp:64-66 */
void ora_client_apply(uint8_t *frame, const int32_t *xs, const uint8_t *diff, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) frame[xs[i]] += diff[i];
}

/* ------------------------------------------------------------------------------------------ */
/* a-5  tests/algorithms_benchmarks.cu                                                          */

void ora_generate_image(int32_t *image, int h, int w, unsigned seed) {
    srand(seed); /* upstream never seeds (== srand(1) for the first image) */
    for (long i = 0; i < (long)h * w * 3; i++) image[i] = rand() % 255; /* :6-8 */
}

void ora_int_diff(const int32_t *cur, const int32_t *prev, int32_t *diff, size_t n) {
    for (size_t i = 0; i < n; i++) diff[i] = cur[i] - prev[i]; /* :27-29 */
}

int ora_check_difference(const int32_t *f1, const int32_t *f2, const int32_t *d, int h, int w) {
    for (int i = 0; i < h; i++)              /* :13 */
        for (int a = 0; a < w; a++)          /* :14 */
            if (f1[i * h + a] - f2[i * h + a] != d[i * h + a]) return -1; /* :15-17 */
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* a-6  noise filter                                                                            */

void ora_gaussian_kernel(float *k, int K, float sigma) {
    float sum = 0;                                               /* server.cpp:21 */
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < K; j++) {
            float x = i - (K - 1) / 2.0;                         /* :24 */
            float y = j - (K - 1) / 2.0;                         /* :25 */
            k[i * K + j] = (1.0 / (2.0 * M_PI * sigma * sigma)) *
                           exp(-((x * x + y * y) / (2.0 * sigma * sigma))); /* :26 */
            sum += k[i * K + j];                                 /* :27 */
        }
    }
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) k[i * K + j] /= sum;         /* :33 */
}

/* tests/noise_filter_benchmark/v2.cu:116-124 computeMeanKernel: every tap 1.0/(K*K), rounded to float */
void ora_mean_kernel(float *k, int K) {
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) k[i * K + j] = 1.0 / (K * K);
}

/* The K x K filter of the reference's filter study, tests/noise_filter_benchmark/v2.cu:36-80 (the same kernel as
 * server/src/kernels.cu:97-136 with K a compile-time constant 3..9, even K included): output (row_o, col_o) sums
 * dev_k[i*K+j] * in[row_o - K/2 + i][col_o - K/2 + j] (v2.cu:43-44,66-70), zero outside the image (:46-54; the
 * halo's third channel is left uninitialised there: not reproduced), float accumulator, i-major / j-minor, one
 * multiply then one add per tap, float -> uint8_t as ora_conv3x3. */
void ora_conv_kxk(const uint8_t *in, uint8_t *out, int w, int h, const float *k, int K) {
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            for (int c = 0; c < 3; c++) {
                float acc = 0.0f;
                for (int i = 0; i < K; i++) {
                    for (int j = 0; j < K; j++) {
                        int yy = y - K / 2 + i, xx = x - K / 2 + j;
                        float px = (yy >= 0 && yy < h && xx >= 0 && xx < w)
                                       ? (float)in[((size_t)yy * w + xx) * 3 + c] : 0.0f;
                        float prod = k[i * K + j] * px;
                        acc = acc + prod;
                    }
                }
                out[((size_t)y * w + x) * 3 + c] = !(acc > 0.0f) ? 0 : acc >= 255.0f ? 255 : (uint8_t)acc;
            }
        }
    }
}

void ora_conv3x3(const uint8_t *in, uint8_t *out, int w, int h, const float *k) {
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            for (int c = 0; c < 3; c++) {
                float acc = 0.0f;                                /* kernels.cu:120-122 */
                for (int i = 0; i < 3; i++) {                    /* :124 */
                    for (int j = 0; j < 3; j++) {                /* :125 */
                        int yy = y + i - 1, xx = x + j - 1;      /* :104-105 */
                        /* zero halo, kernels.cu:107-115 */
                        float px = (yy >= 0 && yy < h && xx >= 0 && xx < w)
                                       ? (float)in[((size_t)yy * w + xx) * 3 + c] : 0.0f;
                        float prod = k[i * 3 + j] * px;          /* :126-128, multiply ... */
                        acc = acc + prod;                        /* ... then add (no FMA) */
                    }
                }
                /* :131-133 float -> uint8_t: truncation toward zero; what does not fit saturates (the device
                 * conversion the reference's store compiles to clamps, a plain C cast would be undefined) */
                out[((size_t)y * w + x) * 3 + c] = !(acc > 0.0f) ? 0 : acc >= 255.0f ? 255 : (uint8_t)acc;
            }
        }
    }
}

/* tests/noise_filter_benchmark/v3.cu:32-47: the reference's swap sort, then the middle element. */
static uint8_t ora_median_of(uint8_t *array, int n) {
    int swapped = 1;
    for (int a = 0; a < n && swapped; a++) {                     /* :34 */
        swapped = 0;
        for (int i = 0; i < n - 1; i++) {                        /* :36 */
            if (array[i] > array[i + 1]) {                       /* :37 */
                uint8_t tmp = array[i];
                array[i] = array[i + 1];
                array[i + 1] = tmp;
                swapped = 1;
            }
        }
    }
    return array[n / 2];                                         /* :46 */
}

void ora_median5x5(const uint8_t *in, uint8_t *out, int w, int h) {
    /* tests/noise_filter_benchmark/v3.cu:49-90 (K = 5): per channel, the median of the 5x5 neighbourhood
     * with zeros outside the image (:61-69; the kernel leaves channel 2 of its halo unset, :68 writes
     * channel 1 twice -- zero is what the code means and what is restated). */
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            for (int c = 0; c < 3; c++) {
                uint8_t win[25];
                int n = 0;
                for (int i = 0; i < 5; i++) {                    /* :79 */
                    for (int j = 0; j < 5; j++) {                /* :80 */
                        int yy = y + i - 2, xx = x + j - 2;      /* :58-59, K/2 = 2 */
                        win[n++] = (yy >= 0 && yy < h && xx >= 0 && xx < w)
                                       ? in[((size_t)yy * w + xx) * 3 + c] : 0;
                    }
                }
                out[((size_t)y * w + x) * 3 + c] = ora_median_of(win, 25);   /* :86-88 */
            }
        }
    }
}

void ora_conv3x3_intacc(const int32_t *in, int32_t *out, int w, int h, const float *k) {
    /* tests/noise_filter_benchmark/cpu.cu:72-98 with K=3 */
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) {
            for (int color = 0; color < 3; color++) {
                int ver = i - 1;
                int sum = 0;
                for (int c = 0; c < 3; c++) {
                    int hor = j - 1;
                    for (int d = 0; d < 3; d++) {
                        if (hor >= 0 && hor < w && ver >= 0 && ver < h)
                            sum += in[(hor + ver * w) * 3 + color] * k[c * 3 + d]; /* :82 */
                        hor++;
                    }
                    ver++;
                }
                out[(i * w + j) * 3 + color] = sum;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a-7  heat map, tests/heat_map_benchmark/cpu.cu:19-27                                         */

static double clamp255(double v) { /* min(max(v,0.0),255.0), cpu.cu:22-24 */
    double m = v > 0.0 ? v : 0.0;
    return m < 255.0 ? m : 255.0;
}

void ora_heat_lut(uint8_t *lut) {
    for (int diff = 0; diff <= 765; diff++) {
        float diff1 = diff / (255.0 * 2.0);                                    /* :21 */
        int r = clamp255(sin(M_PI * diff1 - M_PI / 2.0) * 255.0);              /* :22 */
        int g = clamp255(sin(M_PI * diff1) * 255.0);                           /* :23 */
        int b = clamp255(sin(M_PI * diff1 + M_PI / 2.0) * 255.0);              /* :24 */
        lut[diff * 3 + 0] = (uint8_t)b;                                        /* :62 */
        lut[diff * 3 + 1] = (uint8_t)g;                                        /* :63 */
        lut[diff * 3 + 2] = (uint8_t)r;                                        /* :64 */
    }
}

void ora_heat_map(const uint8_t *cur, const uint8_t *prev, uint8_t *out, size_t npix) {
    uint8_t lut[766 * 3];
    ora_heat_lut(lut);
    for (size_t p = 0; p < npix; p++) {
        int d = abs((int)cur[3 * p] - (int)prev[3 * p]) +
                abs((int)cur[3 * p + 1] - (int)prev[3 * p + 1]) +
                abs((int)cur[3 * p + 2] - (int)prev[3 * p + 2]);               /* :60 */
        out[3 * p] = lut[d * 3];
        out[3 * p + 1] = lut[d * 3 + 1];
        out[3 * p + 2] = lut[d * 3 + 2];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a-10  red motion map                                                                         */

void ora_red_dense(const uint8_t *cur, const uint8_t *prev, uint8_t *out, size_t npix, int thr) {
    for (size_t p = 0; p < npix; p++) { /* tests/heat_map_red_benchmark/cpu.cu:38-55 */
        int f = abs((int)cur[3 * p] - (int)prev[3 * p]) > thr ||
                abs((int)cur[3 * p + 1] - (int)prev[3 * p + 1]) > thr ||
                abs((int)cur[3 * p + 2] - (int)prev[3 * p + 2]) > thr;       /* :44 */
        out[3 * p] = 0;
        out[3 * p + 1] = 0;
        out[3 * p + 2] = f ? 255 : 0;
    }
}

void ora_red_overlap(uint8_t *img, const int32_t *xs, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) img[xs[i] + (2 - xs[i] % 3)] = 255; /* kernels.cu:279 */
}

/* ------------------------------------------------------------------------------------------ */
/* a-8  grayscale                                                                               */

void ora_gray_avg(const uint8_t *in, uint8_t *out, size_t npix) {
    for (size_t p = 0; p < npix; p++) { /* server.cpp:96-101 */
        int sum = in[3 * p] + in[3 * p + 1] + in[3 * p + 2];
        out[3 * p] = sum / 3;
        out[3 * p + 1] = sum / 3;
        out[3 * p + 2] = sum / 3;
    }
}

uint8_t ora_gray_weighted_px(uint8_t b, uint8_t g, uint8_t r) {
    /* tests/grayscale-weighted/cpu.cu:40, evaluated in double left to right */
    double v = 0.114 * b + 0.587 * g + 0.299 * r;
    return (uint8_t)v;
}

void ora_gray_weighted(const uint8_t *in, uint8_t *out, size_t npix) {
    for (size_t p = 0; p < npix; p++) {
        uint8_t v = ora_gray_weighted_px(in[3 * p], in[3 * p + 1], in[3 * p + 2]);
        out[3 * p] = v; out[3 * p + 1] = v; out[3 * p + 2] = v; /* kernels.cu:88-90 layout */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a-9  binarize chain, server/src/server.cpp:103-135                                           */

void ora_histogram(const uint8_t *gray3, size_t nbytes, int32_t *hist) {
    memset(hist, 0, 256 * sizeof *hist);                         /* :103 */
    for (size_t i = 0; i < nbytes; i += 3) hist[gray3[i]]++;     /* :104-106 */
}

int ora_two_max_threshold(const int32_t *histogram) {
    int max = -1, sec_max = -1;                                  /* :108 */
    int index_max = -1, index_sec_max = -1;                      /* :109 */
    for (int i = 0; i < 256; i++) {
        if (histogram[i] >= max) {                               /* :111 */
            index_sec_max = index_max;
            index_max = i;
            max = histogram[i];
            sec_max = max;
        } else if (histogram[i] > sec_max && histogram[i] < max) { /* :116 (dead) */
            sec_max = histogram[i];
            index_sec_max = i;
        }
    }
    int threshold = (index_max + index_sec_max) / 2;             /* :121 */
    if (threshold < 50) threshold = 50;                          /* :122-124 */
    if (threshold > 200) threshold = 200;                        /* :125-127 */
    return threshold;
}

void ora_binarize(const uint8_t *in, uint8_t *out, size_t nbytes, int thr) {
    for (size_t i = 0; i < nbytes; i++) out[i] = in[i] > thr ? 255 : 0; /* :129-135 */
}

int ora_server_cpu_branch(uint8_t *frame, size_t nbytes) {
    int32_t hist[256];
    ora_gray_avg(frame, frame, nbytes / 3);
    ora_histogram(frame, nbytes, hist);
    int thr = ora_two_max_threshold(hist);
    ora_binarize(frame, frame, nbytes, thr);
    return thr;
}
