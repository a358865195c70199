// internal.h -- shared declarations of libmi355diff (not part of the C-ABI).
#ifndef MI355_INTERNAL_H_
#define MI355_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mi355 {

// Geometry of the diff/threshold/pack kernel: one wave64 owns one 1 KiB *tile* of the frame
// (64 lanes x 16 B, a single global_load_dwordx4 per frame) for the whole batch.
constexpr uint32_t kTileBytes = 1024;
constexpr uint32_t kWavesPerBlock = 4;

struct PackArgs {
    const uint8_t *cur;    // frame t at cur + t*stride
    const uint8_t *prev;   // pair mode: prev frame t at prev + t*stride; stream mode: unused
    uint8_t *state;        // stream mode: N bytes, read at start and written back at the end
    size_t stride;         // bytes between frames
    uint32_t n;            // bytes per frame
    int32_t nframes;       // T
    int32_t thr;           // threshold
    uint32_t ntiles;       // W = ceil(n / 1024)
    uint32_t tile_begin;   // this launch packs tiles [tile_begin, tile_end): the whole frame, or one part of it when a
    uint32_t tile_end;     // pipelined batch is packed by two staggered launches (core.hip, MI355_OPT_SPLIT_PCT)
    uint32_t *codes;       // code log: T/4 chunks x W tiles x 256 codes (one per candidate lane)
    uint4 *rec;            // record log: T chunks x W tiles x 64 records of 16 masked diff bytes (multi-byte lanes)
    uint4 *meta;           // [T][W]: {code position, record position, flagged bytes, candidates | multi-byte lanes << 16}
    uint32_t codes_bytes;  // sizes of the logs (buffer descriptors; all < 2^32)
    uint32_t rec_bytes;
    uint32_t meta_bytes;
};

struct ExpandArgs {
    const uint32_t *codes;
    const uint4 *rec;
    const uint4 *meta;        // [T][W]
    const uint32_t *roff;     // [T][4G] flagged bytes of the frame before each range of 16 tiles (G = ceil(W/64) groups)
    const uint32_t *offsets;  // [T+1]   exclusive scan of the frame totals
    uint32_t ntiles;
    uint32_t codes_bytes;     // sizes of the two logs (the expander's buffer descriptors; both < 2^32)
    uint32_t rec_bytes;
    int32_t *out_xs;
    uint8_t *out_diff;
    uint8_t *wire;            // != nullptr: write the sender's byte stream here instead of out_xs/out_diff
    size_t capacity;          // entries of out_xs/out_diff, or bytes of wire
};

// core.hip, for the other translation units of the library (group.hip)
}  // namespace mi355
struct mi355_core;
namespace mi355 {
int set_error(int code, const char *what);        // sets the calling thread's mi355_last_error text, returns code
hipStream_t core_stream(::mi355_core *c);   // the stream the core currently enqueues on (joins a pipelined batch's side stream first)
int core_device(const ::mi355_core *c);

// diff_pack.hip
hipError_t launch_diff_pack(const PackArgs &a, bool pair, bool aligned, bool pair_once /* pair mode: no frame is an operand twice */,
                            uint32_t max_blocks /* 0: one tile per wave */, hipStream_t s);
uint32_t expand_groups(uint32_t ntiles);
// A frame total travels as ONE 64-bit word {total: 31 bits (a frame is below 2 GiB), tag of the launch: 33 bits}
// (diff_pack.hip, publish_total).  The tag counts the launches of a core and is never 0 (0 = never written); when it
// would wrap the host clears every total behind a synchronisation (core.hip, next_scan_epoch), so a slot can never
// carry a stale word with the current tag, whatever the number of launches.
constexpr int kTotalBits = 31;
constexpr uint64_t kEpochWrap = 1ull << (64 - kTotalBits);   // tags are 1 .. kEpochWrap - 1
// the tag of the next launch; true: the tags have wrapped, the caller must clear the totals before that launch
inline bool next_scan_epoch(uint64_t &epoch) {
    if (++epoch < kEpochWrap) return false;
    epoch = 1;
    return true;
}
hipError_t launch_scan(const uint4 *meta, uint32_t *roff, uint64_t *totals /* [T] {total, tag} */, uint32_t ntiles,
                       int nframes, uint32_t *offsets, uint32_t *ticket /* zero between launches */,
                       uint64_t epoch /* 1 .. kEpochWrap - 1, different from every tag `totals` still holds */,
                       uint64_t *note /* pinned host word for {batch total, frames << 32}, or nullptr */, hipStream_t s);
hipError_t launch_expand(const ExpandArgs &a, int nframes, hipStream_t s);

// stream_ops.hip
constexpr int kMaxParts = 64;
struct MergeArgs {
    const uint32_t *part_off;   // [nparts][T+1] device: each part's own exclusive scan
    const int32_t *xs_all;      // parts' entries back to back, part p at part_base[p]
    const uint8_t *diff_all;
    int32_t *out_xs;
    uint8_t *out_diff;
    size_t capacity;
    int32_t nparts, nframes;
    uint32_t part_base[kMaxParts];
    int32_t xs_bias[kMaxParts];
};
hipError_t launch_apply(uint8_t *frame, uint32_t nbytes, const void *xs, const void *diff,
                        const uint32_t *d_offsets, int t, uint32_t host_count, hipStream_t s);
hipError_t launch_apply_all(uint8_t *frame, uint32_t nbytes, const int32_t *xs, const uint8_t *diff,
                            const uint32_t *d_offsets, int nframes, hipStream_t s);
hipError_t launch_export(const uint32_t *offsets, const int32_t *xs, const uint8_t *diff, int32_t *h_xs,
                         uint8_t *h_diff, uint32_t *h_count, hipStream_t s);
hipError_t launch_merge(const MergeArgs &a, uint32_t *out_offsets, hipStream_t s);

// filters.hip -- every per-frame kernel takes a FrameBatch: frame f lives at base + f*stride
struct FrameBatch {
    size_t stride;
    int nframes;
};
hipError_t launch_int_diff(const int32_t *cur, const int32_t *prev, int32_t *out, size_t n,
                           hipStream_t s);
hipError_t init_gray_table();   // the weighted gray's exception table on the current device
hipError_t launch_gray(const uint8_t *in, uint8_t *out, uint32_t npix, bool weighted, FrameBatch fb,
                       hipStream_t s);
hipError_t launch_binarize_chain(const uint8_t *gray, uint8_t *out, uint32_t nbytes, int32_t *hist,
                                 int32_t *thr, FrameBatch fb, hipStream_t s);
hipError_t launch_gray_binarize_fused(const uint8_t *color, uint8_t *out, uint32_t npix, bool weighted,
                                      int32_t *hist, int32_t *thr, FrameBatch fb, hipStream_t s,
                                      uint8_t *gray1 /* scratch: nframes x gray1_stride bytes, or nullptr */,
                                      size_t gray1_stride /* >= npix, a multiple of 16 */);
hipError_t launch_heat_map(const uint8_t *cur, const uint8_t *prev, uint8_t *out, uint32_t npix,
                           const uint8_t *lut, FrameBatch fb, hipStream_t s);
hipError_t launch_red_dense(const uint8_t *cur, const uint8_t *prev, uint8_t *out, uint32_t npix,
                            int thr, FrameBatch fb, hipStream_t s);
hipError_t launch_red_overlap(uint8_t *img, const int32_t *xs, const uint32_t *d_count,
                              uint32_t count, uint32_t nbytes, hipStream_t s);
uint32_t red_bounds_per_frame(uint32_t nbytes);
hipError_t launch_red_stream(uint8_t *out, const uint32_t *offsets, const int32_t *xs, uint32_t nbytes, bool clear,
                             FrameBatch fb, hipStream_t s,
                             uint32_t *bounds_scratch /* nframes x red_bounds_per_frame(nbytes) words, or nullptr */);
hipError_t launch_conv3x3(const uint8_t *in, uint8_t *out, int w, int h, const float *k9, bool k9_symmetric,
                          FrameBatch fb, hipStream_t s);
hipError_t launch_conv_kxk(const uint8_t *in, uint8_t *out, int w, int h, const float *kk /* device, K*K */, int K,
                           FrameBatch fb, hipStream_t s);
hipError_t launch_median5x5(const uint8_t *in, uint8_t *out, int w, int h, int rows_per_band /* 0: chosen here */, FrameBatch fb,
                            hipStream_t s);
hipError_t launch_blit_glyph(uint8_t *frame, const uint8_t *glyph, int glyph_h, int glyph_wbytes,
                             int x_off_bytes, int frame_wbytes, int frame_h, hipStream_t s);

}  // namespace mi355
#endif
