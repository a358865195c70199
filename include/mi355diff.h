/*
 * mi355diff.h -- C-ABI of libmi355diff.so: the MI355X (gfx950) frame-differencing + filter core.
 *
 * This is the drop-in boundary for the hot path of MatteoBattilana/CUDAVideoStream: everything the
 * reference's `diff::cuda::CUDACore` (server/include/kernels.cuh:13-43, server/src/kernels.cu:377-536)
 * does on the GPU, behind plain C entry points (no C++/STL/torch types).  The C++ class of the same
 * name that the reference's server.cpp links against lives in cudavideostream_amd/compat/ and only
 * forwards to these functions; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - A frame is width*height BGR24 pixels, row-major, N = 3*width*height bytes
 *     (server/src/server.cpp:46).  All arithmetic is per byte, as in the reference.
 *   - "d_" arguments are device (HBM) pointers of the core's device; everything else is host memory.
 *   - One core = one device = one stream = one caller thread at a time (the reference calls exec_core
 *     from a single thread, server/src/server.cpp:139).  Several cores (one per GPU) are independent.
 *   - Every function returns MI355_OK (0) or a negative error; mi355_last_error() gives the text for
 *     the calling thread.  Device-resident entry points are asynchronous on the core's stream and
 *     never synchronise; host-buffer entry points return after the results are in the host buffers.
 *   - There is no CPU fallback: without a usable HIP device mi355_create() fails.
 */
#ifndef MI355DIFF_H_
#define MI355DIFF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_ERR_INVALID (-1)  /* bad argument / configuration */
#define MI355_ERR_HIP (-2)      /* a HIP runtime call failed */
#define MI355_ERR_STATE (-3)    /* call not valid in the core's current state */

/* Visualiser selection of exec(): the values of NOISE_VISUALIZER in server/include/common.h:11. */
#define MI355_VIS_NONE 0
#define MI355_VIS_HEAT 1         /* kernels.cu:480  heat_map                      */
#define MI355_VIS_RED 2          /* kernels.cu:513  memset + red_black_map_overlap */
#define MI355_VIS_RED_OVERLAP 3  /* kernels.cu:517  red on the previous frame      */
#define MI355_VIS_GRAY 4         /* kernels.cu:487  grayscale_kernel_v3 (weighted) */
#define MI355_VIS_BINARIZE 5     /* kernels.cu:493-498 gray + histogram + max + binarize */

typedef struct mi355_core mi355_core;

typedef struct mi355_config {
    int32_t width;      /* pixels */
    int32_t height;     /* pixels */
    int32_t threshold;  /* LR_THRESHOLDS, server/include/common.h:14 (20); strict >, 0..255 (255 flags nothing) */
    int32_t max_batch;  /* largest nframes of a *_batch call; sizes the workspace (>= 1) */
    int32_t device;     /* HIP device ordinal, or -1 for the current device */
    int32_t noise_filter; /* != 0: exec() runs the 3x3 convolution first (NOISE_FILTER, common.h:5) */
    int32_t visualizer; /* MI355_VIS_* used by exec() (NOISE_VISUALIZER, common.h:11) */
    int32_t flags;      /* MI355_FLAG_* (below), or 0; unknown bits are refused */
} mi355_config;
/* MI355_FLAG_OWN_QUEUES: the core's streams (its own stream and the two side streams of pipelined batches) are created in the
 * LEAST stream-priority class.  HIP serves the streams of one priority class of a process with at most GPU_MAX_HW_QUEUES (4)
 * hardware queues; a process that also holds a framework's stream pools in the default class -- PyTorch creates 32 per class
 * the moment torch.distributed / RCCL asks for one -- leaves the core's three streams sharing queues, and kernels that are
 * meant to run BESIDE each other (the expansion of batch k and the pack kernel of batch k + 1) run one after the other:
 * 462 k instead of 560-620 k frames/s at 1080p (round 6, profiles/README.md r06k / r06l).  No framework uses the least class,
 * so there the three streams get hardware queues of their own (545-551 k with or without torch.distributed in the
 * process; in a process WITHOUT other streams the default class is 1-2 % faster, hence a flag).  Set it in any process that
 * also runs PyTorch collectives / RCCL / other HIP streams (bench.py does under a launcher). */
#define MI355_FLAG_OWN_QUEUES 1

/* ABI version of this header: bumped whenever an existing entry point changes its argument list (round 3 did that to
 * mi355_group_gather without a marker: a caller built against the older header still linked and passed shifted
 * arguments).  A binding checks mi355_abi_version() == MI355_ABI_VERSION when it loads the library
 * (cudavideostream_amd/lib.py, compat/include/group.hpp do).
 *   3  round 3: mi355_group_gather gained member_capacity (6th argument)
 *   4  round 4: + mi355_abi_version, mi355_probe_clock, mi355_probe_hbm_read (additions only)
 *   5  round 5: + mi355_set_option / mi355_get_option; MI355_FLAG_FUSED / MI355_FLAG_CHAIN (two opt-in experiments) are
 *      gone and cfg.flags must be 0; the environment variables MI355_SPLIT, MI355_DENSE_PCT, MI355_CHAIN_HINT and the
 *      undocumented tuning variables are no longer read (options below); + MI355_OPT_MEDIAN_ROWS, mi355_probe_hbm_write
 *      (additions)
 *   6  round 6: + mi355_prepare, mi355_alloc_outputs, MI355_OPT_SCAN_EPOCH_LEFT, cfg.flags bit MI355_FLAG_OWN_QUEUES (additions only) */
#define MI355_ABI_VERSION 6
int mi355_abi_version(void);

/* ---- life cycle: CUDACore::CUDACore (kernels.cu:377-428) without the uploads ------------------ */
int mi355_create(const mi355_config *cfg, mi355_core **out);
void mi355_destroy(mi355_core *core);
const char *mi355_last_error(void);
/* Size in bytes of one frame (3*width*height). */
size_t mi355_frame_bytes(const mi355_core *core);
/* Bytes of HBM workspace held by the core (state, logs, counters).  The logs are sized for the worst case (every byte
 * of every frame of a batch changed): about (1 + 1/3) * max_batch * N bytes, 2.07 GB for 1080p and max_batch 256.  The
 * first asynchronous batch call on the core's own stream allocates a second set (hipMalloc + hipMemset inside that
 * call: it is not asynchronous) for the pipelined mode below; scratch buffers of the fused gray+binarize chain and of
 * the cleared red map are likewise allocated by the first call that needs them -- unless mi355_prepare has made them. */
size_t mi355_workspace_bytes(const mi355_core *core);
/* Makes NOW what the entry points would otherwise make on first use -- so that no hipMalloc / hipMemset / stream or event
 * creation happens inside an asynchronous entry point and a server's first frames do not stall.  `what` is a mask:
 *   MI355_PREPARE_BATCHES      second set of logs, side streams and events of pipelined own-stream batches
 *                              (mi355_diff_stream_batch / _pairs_batch / _wire_batch on the core's own stream);
 *   MI355_PREPARE_GRAY_CHAIN   one gray byte per pixel for max_batch frames (MI355_OP_GRAY_*_BINARIZE, MI355_VIS_BINARIZE);
 *   MI355_PREPARE_RED_CLEAR    slice bounds of mi355_red_stream_batch(clear != 0);
 *   MI355_PREPARE_CONV_KXK     the tap buffer of mi355_conv_kxk;
 *   MI355_PREPARE_EXEC         one pass of mi355_exec's own kernels (noise filter if its kernel is set, visualiser, pack,
 *                              index, expansion, red map, export) over a copy of the CURRENT state -- nothing differs, so
 *                              the state and the caller's buffers stay as they are -- so that the first real frame does
 *                              not pay for the first use of each kernel (measured: the first exec_core of a fresh core
 *                              1.4 ms, the later ones 0.16).  The C++ drop-in's constructor calls it.
 * mi355_create already makes the one a per-frame server needs (MI355_VIS_BINARIZE's gray bytes).  Blocking; idempotent;
 * mi355_workspace_bytes grows by what was made.  After mi355_prepare(core, MI355_PREPARE_ALL) no entry point of the core
 * allocates (mi355_pipe_open and mi355_set_glyphs, which say so, excepted). */
#define MI355_PREPARE_BATCHES 1u
#define MI355_PREPARE_GRAY_CHAIN 2u
#define MI355_PREPARE_RED_CLEAR 4u
#define MI355_PREPARE_CONV_KXK 8u
#define MI355_PREPARE_EXEC 16u
#define MI355_PREPARE_ALL 31u
int mi355_prepare(mi355_core *core, unsigned what);

/* Streams.  A core starts on a stream of its own, created hipStreamNonBlocking: it is NOT ordered against the
 * legacy default stream nor against any other stream of the caller.  Buffers the caller fills asynchronously on
 * another stream (hipMemcpyAsync, a framework's kernels) must be complete -- hipStreamSynchronize / an event the
 * caller waits for -- before an asynchronous entry point of the core reads them, and the caller must
 * mi355_synchronize (or use the blocking entry points) before it reads the outputs on another stream.
 * mi355_set_stream makes the core enqueue on an existing hipStream_t instead (e.g. PyTorch's current stream; NULL is
 * the default stream), so that the caller's own work on that stream is ordered with the core's; mi355_use_own_stream
 * goes back.  Both wait for the work already queued on the stream being left (all batches of a core share one
 * workspace). */
int mi355_set_stream(mi355_core *core, void *hip_stream);
int mi355_use_own_stream(mi355_core *core);
int mi355_synchronize(mi355_core *core);
/* Lifetime of the caller's buffers.  The device-resident entry points are asynchronous: every d_ buffer handed to one
 * (inputs AND outputs) must stay allocated, and the inputs unmodified, until the work has completed -- mi355_synchronize,
 * or, with mi355_set_stream, the caller's own synchronisation of that stream.  A framework whose allocator recycles a
 * freed tensor for other kernels on ITS stream (PyTorch's caching allocator) must keep the tensors referenced until
 * then: on the core's own stream the library's kernels are not ordered against the framework's stream.
 * (cudavideostream_amd/core.py holds such references itself until synchronize().) */

/* Options.  The schedule of the batches on the core's own stream can be tuned per core; RESULTS never depend on it.
 * Changing an option first waits for the work the core has queued.  Of the environment the library reads two variables
 * and nothing else: MI355_PIPELINE=0 makes MI355_OPT_PIPELINE default to 0 for every core of the process (a switch for
 * the operator of an unmodified server binary); and, in the multi-GPU entry points only, MI355_RCCL_LIB=path names
 * another library with the ten RCCL entry points they bind instead of librccl.so.1 (the tests' stand-ins, which let
 * several ranks share the one GPU of a test box). */
#define MI355_OPT_PIPELINE 1     /* 1 (default): own-stream batches are pipelined (below); 0: one kernel after the other */
#define MI355_OPT_SPLIT_PCT 2    /* 50 (default): a pipelined batch is packed by two launches, this share of the tiles on
                                  * the first; 5..95, or 0 = one launch */
#define MI355_OPT_DENSE_PCT 3    /* 40 (default): batches in which more than this share of the bytes changed are not
                                  * overlapped (adaptive overlap, below); 0 = always overlap, 100 = same */
#define MI355_OPT_CHAIN_HINT 4   /* 1 (default): a batch that follows a frame filter on this core is not overlapped; 0: is */
#define MI355_OPT_PACK_BLOCKS 5  /* -1 (default): the pipelined pack kernel runs on 4 workgroups per CU; 0: one tile per
                                  * wave; n > 0: n workgroups */
#define MI355_OPT_MEDIAN_ROWS 6  /* 0 (default): the 5x5 median's column-strip kernel walks bands of 5..60 rows, chosen per launch
                                  * (from 20 rows up the length that wastes least of the frame's last pair of bands; shorter
                                  * when that makes fewer than a few thousand waves); 5, 10, .. 60: this many */
#define MI355_OPT_SCAN_EPOCH_LEFT 7 /* tests only: launches of the index kernel left before its 33-bit launch tag wraps (the
                                  * totals are then cleared behind a synchronisation and the tag restarts at 1); set: 1..2^30 */
int mi355_set_option(mi355_core *core, int option, int value);
int mi355_get_option(mi355_core *core, int option, int *value);

/* ---- state: the reconstructed client frame ("previous" with negative feedback) ------------------
 * kernels.cu:406 uploads the base frame into d_current; after each frame the surviving buffer is
 * cur where |df| > threshold and prev elsewhere (kernels.cu:312-331, tests/cuda_streaming/test.cu:571).
 * The core keeps ONE persistent state buffer with exactly those contents. */
int mi355_set_state(mi355_core *core, const uint8_t *host_frame);
int mi355_get_state(mi355_core *core, uint8_t *host_frame);
void *mi355_state_device_ptr(mi355_core *core);

/* ---- constants: cudaMemcpyToSymbol(dev_k) kernels.cu:394, glyph upload kernels.cu:381-382 ------ */
int mi355_set_conv_kernel(mi355_core *core, const float *k9);
int mi355_set_glyphs(mi355_core *core, const uint8_t *chars_px, int nglyphs, int glyph_h, int glyph_w,
                     const char *charset);

/* ---- the hot path, device resident ---------------------------------------------------------------
 * kernel2 (kernels.cu:289-334) over a batch.  For every frame t (in order) and every byte i
 * (ascending): df = frame[t][i] - state[i]; if |df| > threshold emit (xs = i, diff = (uint8)df) and
 * state[i] = frame[t][i].  Output is the CPU path's order (tests/cuda_streaming/test.cu:563-573),
 * not the reference kernel's atomicInc order.
 *   d_frames : nframes frames, frame t at d_frames + t*stride_bytes (stride_bytes >= N; the fast
 *              path needs d_frames and stride_bytes to be multiples of 16)
 *   d_offsets: uint32[nframes+1], exclusive scan of the per-frame counts (offsets[0] = 0)
 *   d_xs     : int32[capacity]  byte indices, frame t's entries at [offsets[t], offsets[t+1])
 *   d_diff   : uint8[capacity]  (uint8)df of the same entries
 * Entries beyond `capacity` are dropped (offsets stay exact), so check offsets[nframes] <= capacity.
 * Asynchronous.  On the core's OWN stream consecutive batches are pipelined: the index and the expansion of a
 * batch run on a side stream beside the next batch's pack kernel.  A batch's outputs (d_offsets, d_xs, d_diff,
 * d_wire) are complete after mi355_synchronize and for every later call on this core that can consume them
 * (mi355_apply_*, mi355_red_stream_batch / _red_overlap, mi355_merge_parts, mi355_download, mi355_exec /
 * mi355_pipe_*, the group gather) -- these first wait for the last expansion.  The frame filters (mi355_filter_batch,
 * mi355_gray_*, mi355_binarize_chain, mi355_heat_map, mi355_red_dense, mi355_conv*, mi355_median5x5,
 * mi355_int_diff) take frames, not packed streams, and are ordered on the core's stream only: they may run beside
 * the expansion of the batch before (visualiser of frame k + 1 beside the expansion of frame k).  With a caller's
 * stream (mi355_set_stream) nothing is pipelined: every kernel runs on that stream, in call order.
 * MI355_OPT_PIPELINE 0 (or MI355_PIPELINE=0 in the environment) switches the pipelining off.  (A pipelined batch is packed
 * by two kernel launches on two streams of the core, half the tiles each; MI355_OPT_SPLIT_PCT 0 packs it with one.)
 * A batch that follows a frame filter on this core (the server's visualiser or noise filter in front of every diff) is not
 * overlapped either: one kernel after the other measured faster for such chains (MI355_OPT_CHAIN_HINT 0: overlap
 * regardless).
 * The overlap is adaptive: a batch in which more than 40 % of the bytes changed (a scene change; MI355_OPT_DENSE_PCT) has an
 * expansion longer than its pack kernel and loses by running beside the next batch.  The index kernel of every own-stream
 * batch leaves the batch's total in a word of pinned host memory; the library, without ever waiting for it, runs batches
 * one after the other while the latest total that has arrived says "dense".  Only the schedule depends on it, never a result.
 * Cache policy: the frames of a stream are read, and every output (d_xs, d_diff, d_wire; the visualiser frames of the
 * filters) is written, with non-temporal instructions -- each is touched once.  A consumer that reads the packed stream
 * right behind the batch (mi355_apply_*, the red map, the gather) reads it from memory, not from the caches. */
int mi355_diff_stream_batch(mi355_core *core, const void *d_frames, size_t stride_bytes, int nframes,
                            void *d_offsets, void *d_xs, void *d_diff, size_t capacity);

/* Stateless form (tests/algorithms_benchmarks.cu style frame pairs): frame t is compared with
 * d_prev + t*stride_bytes instead of the state; the core's state is neither read nor written.
 * The two operands may overlap in any way (pairs of consecutive frames of one buffer: d_prev = frames,
 * d_cur = frames + stride).  The library looks at the addresses: operands that share a frame are read through the
 * caches (the second read of a frame hits), operands that share none -- separate buffers, or the pairs (f - 1, f) of
 * every 8th f that a round-robin shard diffs -- with non-temporal loads. */
int mi355_diff_pairs_batch(mi355_core *core, const void *d_cur, const void *d_prev,
                           size_t stride_bytes, int nframes, void *d_offsets, void *d_xs,
                           void *d_diff, size_t capacity);

/* ---- the stream either side of the path ---------------------------------------
 * Wire form of mi355_diff_stream_batch: instead of separate (xs, diff) arrays the batch leaves as the exact
 * byte stream the reference's sender thread writes per frame (server/src/threads.cpp:227-229):
 *     u32 n (= h_pos) | i32 xs[n] | u8 diff[n]
 * frames back to back, frame t at byte 4*t + 5*offsets[t] of d_wire; mi355_wire_bytes(nframes,
 * offsets[nframes]) bytes in all, ready for one write() to the socket after the base frame
 * (threads.cpp:224, mi355_get_state before the first batch).  A frame that does not fit in capacity_bytes
 * is dropped whole (its header is still written when it fits); d_offsets is exact regardless. */
int mi355_diff_stream_wire_batch(mi355_core *core, const void *d_frames, size_t stride_bytes, int nframes,
                                 void *d_offsets, void *d_wire, size_t capacity_bytes);
size_t mi355_wire_bytes(int nframes, uint64_t entries);

/* The client's side, client/opencv.cpp:50-66: for every frame in order, state[xs[i]] += diff[i] (uint8
 * wrap-around) on the core's state (a client core is a core whose state was set to the received base frame,
 * opencv.cpp:38-46).  d_frames_out != NULL: the reconstructed frame t is also copied to d_frames_out +
 * t*stride_bytes (what the client shows, opencv.cpp:68); NULL: only the state advances, all frames in one
 * launch.  Indices >= N are ignored.  Asynchronous on the core's stream. */
int mi355_apply_batch(mi355_core *core, const void *d_offsets, const void *d_xs, const void *d_diff,
                      int nframes, void *d_frames_out, size_t stride_bytes);
/* Same from the wire bytes; h_counts[t] (host) are the headers the client has read from the socket
 * (opencv.cpp:52), the header words inside d_wire are skipped, not trusted. */
int mi355_apply_wire_batch(mi355_core *core, const void *d_wire, const uint32_t *h_counts, int nframes,
                           void *d_frames_out, size_t stride_bytes);

/* Several cores may each own a row band of ONE stream (bands are contiguous byte ranges of the frame, so a
 * band is a core of the band's height fed with d_frames + band_start).  After the
 * bands' streams have been gathered back to back (part p's entries at h_part_base[p], its own index
 * d_part_offsets[p][0..nframes]), this merges them into the single stream of the whole frame: frame t =
 * the parts' frame-t segments in part order, xs + h_xs_bias[p] (the band's first byte). */
int mi355_merge_parts(mi355_core *core, int nparts, int nframes, const void *d_part_offsets,
                      const uint32_t *h_part_base, const int32_t *h_xs_bias, const void *d_xs_all,
                      const void *d_diff_all, void *d_offsets, void *d_xs, void *d_diff, size_t capacity);

/* Integer difference of tests/algorithms_benchmarks.cu:24-30 (kernel1): d[i] = cur[i] - prev[i] on
 * int32 arrays of n elements, no threshold, no pack. */
int mi355_int_diff(mi355_core *core, const void *d_cur, const void *d_prev, void *d_out, size_t n);

/* ---- filters, device resident (all N-byte BGR24 frames; in-place allowed unless noted) --------- */
/* kernels.cu:31-43 / server.cpp:96-101: s = (B+G+R)/3 into the 3 channels. */
int mi355_gray_avg(mi355_core *core, const void *d_in, void *d_out);
/* kernels.cu:67-95 / tests/grayscale-weighted/cpu.cu:40: (uint8)(0.114*B + 0.587*G + 0.299*R). */
int mi355_gray_weighted(mi355_core *core, const void *d_in, void *d_out);
/* kernels.cu:138-241 / server.cpp:103-135: histogram of every 3rd byte, two-max threshold clamped
 * to [50,200] (CPU semantics), binarize.  d_hist (int32[256]) and d_thr (int32[1]) may be NULL. */
int mi355_binarize_chain(mi355_core *core, const void *d_gray, void *d_out, void *d_hist, void *d_thr);
/* kernels.cu:243-270 / tests/heat_map_benchmark/cpu.cu:19-27,54-66. */
int mi355_heat_map(mi355_core *core, const void *d_cur, const void *d_prev, void *d_out);
/* tests/heat_map_red_benchmark/cpu.cu:38-55 (dense red/black map). */
int mi355_red_dense(mi355_core *core, const void *d_cur, const void *d_prev, void *d_out);
/* kernels.cu:273-281: img[x + (2 - x%3)] = 255 for the n indices in d_xs; n is read from d_count
 * (uint32 on the device) when d_count != NULL, else `count` is used. */
int mi355_red_overlap(mi355_core *core, void *d_img, const void *d_xs, const void *d_count,
                      uint32_t count);
/* The same from a packed stream, for a batch: frame t (at d_frames + t*stride_bytes) gets R = 255 for every
 * pixel owning one of its entries xs[offsets[t] .. offsets[t+1]).  clear != 0 zeroes the frames first
 * (NOISE_VISUALIZER 2, kernels.cu:513); clear == 0 paints onto what they hold (NOISE_VISUALIZER 3, :517).
 * clear != 0 builds every frame from its entries in ONE write-only pass and needs them ASCENDING within the
 * frame -- as every diff entry point of this library produces them (the pass finds a slice's entries by
 * binary search; entries out of order would be dropped without an error) -- and nframes <= max_batch.
 * clear == 0 is a plain scatter: any order, any nframes. */
int mi355_red_stream_batch(mi355_core *core, const void *d_offsets, const void *d_xs, int nframes,
                           void *d_frames, size_t stride_bytes, int clear);
/* kernels.cu:97-136: 3x3 convolution with the kernel of mi355_set_conv_kernel; not in-place.  fp32, taps in
 * i-major / j-minor order, one multiply then one add per tap; the float result is truncated toward zero and
 * saturated to [0, 255] (any nine floats are accepted: negative taps and sums above 255 clamp). */
int mi355_conv3x3(mi355_core *core, const void *d_in, void *d_out);
/* The K x K form of the same filter as the reference's filter study runs it, K = 1..9, even K included
 * (tests/noise_filter_benchmark/v2.cu:36-80: taps at rows / columns -K/2 .. K-1-K/2, zero outside the image); k =
 * K*K floats in host memory, row-major (v2.cu:116-124 mean, :139-160 Gaussian).  Same arithmetic and conversion as
 * mi355_conv3x3 (for K = 3 the two agree bit for bit); not in-place; not tuned -- the server's path is K = 3. */
int mi355_conv_kxk(mi355_core *core, const void *d_in, void *d_out, const float *k, int K);

/* tests/noise_filter_benchmark/v3.cu:32-90 (the K = 5 median the reference evaluated and left out of its
 * server for speed): per channel the median of the 5x5 neighbourhood, zeros outside the image; not in-place. */
int mi355_median5x5(mi355_core *core, const void *d_in, void *d_out);

/* Batched form of the per-frame filters: nframes frames at d_in + t*stride_bytes (and d_in2 + t*stride_bytes
 * for the two-input filters) -> d_out + t*stride_bytes, one launch per kernel for the whole batch.
 * The *_BINARIZE ops compute one histogram and one two-max threshold per frame; the fused forms read the colour
 * frame ONCE: pass 1 converts it and keeps one gray byte per pixel in a scratch of the core (max_batch * N/3 bytes,
 * allocated at the first such call) beside the histogram, pass 2 binarizes from that scratch (BASELINE config 3). */
#define MI355_OP_GRAY_AVG 1                /* kernels.cu:31-43                         */
#define MI355_OP_GRAY_WEIGHTED 2           /* kernels.cu:67-95                         */
#define MI355_OP_BINARIZE 3                /* gray3 in: kernels.cu:138-241             */
#define MI355_OP_GRAY_AVG_BINARIZE 4       /* colour in: server.cpp:96-135 in one call */
#define MI355_OP_GRAY_WEIGHTED_BINARIZE 5  /* colour in: kernels.cu:493-498 (visualizer 5) */
#define MI355_OP_HEAT_MAP 6                /* d_in = cur, d_in2 = prev: kernels.cu:243-270 */
#define MI355_OP_RED_DENSE 7               /* d_in = cur, d_in2 = prev: test.cu:142-168 */
#define MI355_OP_CONV3X3 8                 /* kernels.cu:97-136, not in place          */
#define MI355_OP_MEDIAN5X5 9               /* noise_filter_benchmark/v3.cu:32-90, not in place */
int mi355_filter_batch(mi355_core *core, int op, const void *d_in, const void *d_in2, void *d_out,
                       size_t stride_bytes, int nframes);

/* ---- the per-frame host entry point: CUDACore::exec_core (kernels.cu:430-525) -------------------
 * frame_data: in = the captured frame (N bytes), out = diff[0..*h_pos)      (kernels.cu:461,522)
 * show_ready: out = the visualisation frame when cfg.visualizer != 0        (kernels.cu:481-518)
 * text      : overlay string (characters of the glyph charset) or NULL      (kernels.cu:466-476)
 * h_pos     : out = number of changed bytes                                 (kernels.cu:507)
 * h_xs      : out = their byte indices, ascending                           (kernels.cu:523)
 * Returns after both device synchronisations of the reference (kernels.cu:508,524). */
int mi355_exec(mi355_core *core, uint8_t *frame_data, uint8_t *show_ready, const char *text,
               uint32_t *h_pos, int32_t *h_xs);

/* ---- the same entry point, pipelined ----------------------------------------------
 * The reference overlaps capture, elaboration and sending with three threads around a ring of six pinned
 * slots (server/src/threads.cpp:59-106,134-147,166-179) but exec_core itself blocks the host twice per
 * frame (kernels.cu:508,524).  Here the overlap moves below the boundary: mi355_pipe_submit enqueues the
 * upload of frame k on a copy stream, its kernels on the core's stream and its results' way back, and
 * returns; while frame k is packed, frame k+1 crosses PCIe.  The changed-byte count never visits the host
 * in between: a device kernel stores exactly count indices/differences, and the count, through the mapped
 * pinned pointers.  mi355_pipe_wait(ticket) blocks until that frame's outputs are in the caller's
 * buffers and returns h_pos.  Frames are processed in submission order (the state carries over).
 *   - every host buffer must be pinned memory of mi355_host_alloc (alloc_arrays, kernels.cu:531-536) and
 *     stay untouched between submit and wait; argument meaning as mi355_exec;
 *   - depth (1..8) frames may be in flight; submitting into a full ring first completes the oldest frame;
 *   - mi355_exec is refused while the pipe is open. */
int mi355_pipe_open(mi355_core *core, int depth);
int mi355_pipe_submit(mi355_core *core, uint8_t *frame_data, uint8_t *show_ready, const char *text,
                      int32_t *h_xs, int64_t *ticket);
int mi355_pipe_wait(mi355_core *core, int64_t ticket, uint32_t *h_pos);
int mi355_pipe_close(mi355_core *core);

/* ---- pinned host memory: CUDACore::alloc_arrays (kernels.cu:531-536) ---------------------------- */
int mi355_host_alloc(void **out, size_t bytes);
int mi355_host_free(void *p);

/* ---- device memory for hosts without a HIP binding ---------------------------------------------------
 * The device-resident entry points take device pointers.  A caller that links the HIP runtime (or PyTorch)
 * brings its own; a host language that reaches this library only through the C-ABI (cgo, JNI, ctypes ...)
 * uses these: plain hipMalloc/hipFree on the core's device, and copies ordered on the core's stream
 * (mi355_upload returns once the host buffer may be reused, mi355_download once the bytes are there). */
int mi355_dev_alloc(mi355_core *core, void **out, size_t bytes);
int mi355_dev_free(mi355_core *core, void *d_ptr);
/* The two output arrays of the batch entry points -- d_xs (capacity x int32) and d_diff (capacity x uint8) -- as a PAIR whose
 * placement in memory lets the DENSE expansion run at its fast speed.  When most bytes of a frame change (a scene cut, the
 * synthetic worst cases S0 / P = N), the expansion is bound by its stores to these two arrays, and their two streams either
 * overlap in the memory system (205 us per 32 dense 1080p frames) or do not (265): a property of the pair's physical memory
 * that no address shows, the same for the life of the arrays; about one pair in six of plain allocations is the fast kind
 * (round 6, profiles/README.md).  This call allocates the index array, then draws value arrays (at most 32), runs for each
 * a batch of noise frames (max_batch pairs, P = 0.85 N: ~1 ms) through the core's own kernels into the pair, and keeps the
 * first pair on which the expansion moves its bytes at the fast rate; the others are freed; *draws (may be NULL) = value
 * arrays drawn.  Blocking; the core's state and results are untouched (the probe batches are stateless pairs).  Arrays that
 * hold fewer than 8 frames' worth of entries, and callers of sparse streams (a webcam's: the expansion is then bound
 * elsewhere), need none of this: plain allocations.  Free both with mi355_dev_free. */
int mi355_alloc_outputs(mi355_core *core, size_t capacity, void **d_xs, void **d_diff, int *draws);
int mi355_upload(mi355_core *core, void *d_dst, const void *host_src, size_t bytes);
int mi355_download(mi355_core *core, void *host_dst, const void *d_src, size_t bytes);

/* ---- multi-GPU: several devices of one node, RCCL over xGMI only for the final changed-pixel gather ------------
 * The reference runs on one GPU (server/src/kernels.cu:385); this is the sharding BASELINE.json asks of the path
 * (SURVEY.md section 8e).  The data path has no collective: every member runs an independent stream (E1), its
 * share of round-robin frame pairs (E1, BASELINE config 5) or a row band of one stream (E2; merge with
 * mi355_merge_parts on the root's core) through the single-device entry points, side by side.  The one exchange
 * step is mi355_group_gather.
 *
 * A group is either all in this process -- mi355_group_create makes one core per device and an RCCL communicator
 * over them (a C++ server linked against libmi355compat.a) -- or one member per process: every process creates
 * its core as usual and calls mi355_group_adopt_rank with the same 128-byte id (made once by
 * mi355_group_unique_id and handed around by the launcher, e.g. torch.distributed broadcast).  RCCL is bound at
 * the first group call (dlopen of librccl.so.1); single-GPU users never load it.  The same "one caller thread
 * at a time" rule as for a core applies to a group. */
#define MI355_GROUP_ID_BYTES 128
typedef struct mi355_group mi355_group;
int mi355_group_create(const mi355_config *cfg, int ndev, const int *devices /* NULL: 0..ndev-1 */,
                       mi355_group **out);
int mi355_group_unique_id(void *id128);
int mi355_group_adopt_rank(mi355_core *core, int nranks, int rank, const void *id128, mi355_group **out);
void mi355_group_destroy(mi355_group *group);   /* destroys the cores mi355_group_create made, not adopted ones */
int mi355_group_ranks(const mi355_group *group);           /* members in all processes */
int mi355_group_local_members(const mi355_group *group);   /* members in this process: 0 <= i < this */
mi355_core *mi355_group_core(mi355_group *group, int i);   /* member i's core: mi355_set_state etc. */
int mi355_group_rank_of(const mi355_group *group, int i);  /* member i's rank in the group */
/* mi355_diff_stream_batch / mi355_diff_pairs_batch on every local member, pointer arrays indexed by local member;
 * asynchronous on each member's stream. */
int mi355_group_diff_stream_batch(mi355_group *group, const void *const *d_frames, size_t stride_bytes, int nframes,
                                  void *const *d_offsets, void *const *d_xs, void *const *d_diff, size_t capacity);
int mi355_group_diff_pairs_batch(mi355_group *group, const void *const *d_cur, const void *const *d_prev,
                                 size_t stride_bytes, int nframes, void *const *d_offsets, void *const *d_xs,
                                 void *const *d_diff, size_t capacity);
/* Gather-v of the members' batch outputs to rank `root`: collective over ALL ranks of the group (every process
 * calls it with its local members' arrays).  On the root's device: d_root_offsets[rank][0..nframes] = that
 * rank's own exclusive scan, d_root_xs / d_root_diff = the ranks' entries back to back in rank order (rank r at
 * the sum of the counts before it; root_capacity entries).  member_capacity = the entries every local member's
 * d_xs / d_diff hold (the `capacity` its batch ran with).  h_counts[rank] receives every rank's total in every
 * process (one host synchronisation, kernels.cu:507-508 reads its count back the same way); the transfers
 * themselves are asynchronous on the members' streams.  d_root_* / root_capacity are ignored where the root is
 * not local.
 * Failures that concern the whole exchange are decided collectively, BEFORE anything is sent, and reported on
 * every rank alike (MI355_ERR_INVALID): a member whose batch overflowed its buffers (offsets[nframes] >
 * member_capacity: the entries beyond were dropped, there is nothing to send), a gathered total above the root's
 * capacity, missing root buffers.  No rank is left waiting for another. */
int mi355_group_gather(mi355_group *group, int root, int nframes, const void *const *d_offsets,
                       const void *const *d_xs, const void *const *d_diff, size_t member_capacity,
                       void *d_root_offsets, void *d_root_xs, void *d_root_diff, size_t root_capacity,
                       uint64_t *h_counts);
int mi355_group_synchronize(mi355_group *group);

/* ---- measurement ---------------------------------------------------------------------------------
 * With timing on, every *_batch call brackets its kernels with HIP events on the stream they are
 * launched on.  mi355_get_timing synchronises the stream and returns the sums since the last reset:
 * ms_pack = the diff/threshold/pack kernel alone, ms_total = pack + scan + gather. */
int mi355_set_timing(mi355_core *core, int enabled);
int mi355_get_timing(mi355_core *core, double *ms_pack, double *ms_total, int *launches);
/* The same sums per kernel: pack (k_diff_pack; both launches of a split pipelined batch, from the start of the first to
 * the end of whichever ends later), scan (k_scan_groups), expand (k_expand).  One batch after the other their sum is
 * ms_total; pipelined batches wait between their pack kernel and their index (for the stream hop), so ms_total is larger. */
int mi355_get_kernel_timing(mi355_core *core, double *ms_pack, double *ms_scan, double *ms_expand, int *launches);
int mi355_reset_timing(mi355_core *core);
/* Diagnostics: the shader clock (MHz) the device holds under an integer-VALU load of about `milliseconds` ms
 * (1..2000), measured inside a kernel as d(s_memtime) / d(s_memrealtime) x 100 MHz, median over all waves
 * (csrc/diag.hip).  The diff path is bound by instruction issue, so its frames/s follow this clock; boards differ.
 * Runs on the core's stream and returns after it has finished; not part of any data path. */
int mi355_probe_clock(mi355_core *core, int milliseconds, double *shader_mhz);
/* Diagnostics: GB/s of a plain streaming read of a temporary `megabytes` MB buffer (64..16384; best of three passes):
 * what this board's memory system gives a read-only kernel.  The pack kernel is bound by it; boards differ by ~7 %. */
int mi355_probe_hbm_read(mi355_core *core, size_t megabytes, double *gbps);
/* Diagnostics: GB/s of plain streaming WRITES into a temporary buffer (non-temporal, best of three passes): narrow = 0:
 * 16 bytes per lane (whole lines, the filters' outputs); narrow = 1: what a dense expansion emits, a 4-byte index and a
 * 1-byte value per lane to two arrays.  Boards with the same read rate differ by a quarter in this one. */
int mi355_probe_hbm_write(mi355_core *core, size_t megabytes, int narrow, double *gbps);

#ifdef __cplusplus
}
#endif
#endif /* MI355DIFF_H_ */
