// compat/include/group.hpp -- diff::cuda::CUDAGroup: the C++ server's handle on several GPUs of one node.
//
// Not in the reference (its server uses device 0, server/src/kernels.cu:385): this is the C++ face of the
// C-ABI's mi355_group_* entry points (include/mi355diff.h "multi-GPU", csrc/group.hip) for a server linked
// against libmi355compat.a -- one core per device, every device its own stream of frames or its share of
// frame pairs, RCCL over xGMI only for the final changed-pixel gather.  Header-only, host-only C++11; errors
// follow the reference's convention (message on stderr + exit, kernels.cu:11-22).
#ifndef MI355_COMPAT_GROUP_HPP_
#define MI355_COMPAT_GROUP_HPP_

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../../include/mi355diff.h"

namespace diff {
namespace cuda {

class CUDAGroup {
    mi355_group *g_;

    static void check(int rc, const char *what) {
        if (rc != MI355_OK) {
            fprintf(stderr, "%s: %s\n", what, mi355_last_error());
            exit(rc);
        }
    }
    CUDAGroup(const CUDAGroup &);
    CUDAGroup &operator=(const CUDAGroup &);

public:
    // ndev devices (0..ndev-1), frames of c columns x r rows, batches of up to max_batch frames
    CUDAGroup(int ndev, int r, int c, int max_batch, int threshold = 20) : g_(0) {
        // the library loaded at run time must be the one this header describes (argument lists changed in round 3)
        if (mi355_abi_version() != MI355_ABI_VERSION) {
            fprintf(stderr, "libmi355diff: ABI version %d, this program was built for %d\n", mi355_abi_version(), MI355_ABI_VERSION);
            exit(1);
        }
        mi355_config cfg = mi355_config();
        cfg.width = c; cfg.height = r; cfg.threshold = threshold; cfg.max_batch = max_batch; cfg.device = -1;
        check(mi355_group_create(&cfg, ndev, 0, &g_), "mi355_group_create");
    }
    ~CUDAGroup() { mi355_group_destroy(g_); }

    int size() const { return mi355_group_ranks(g_); }
    mi355_core *core(int dev) { return mi355_group_core(g_, dev); }
    // the base frame of device dev's stream (kernels.cu:406)
    void set_state(int dev, const uint8_t *frame) { check(mi355_set_state(core(dev), frame), "mi355_set_state"); }

    // kernel2 (kernels.cu:289-334) over a batch on every device; arrays indexed by device, device pointers
    void diff_stream_batch(const std::vector<const void *> &d_frames, size_t stride, int nframes,
                           const std::vector<void *> &d_offsets, const std::vector<void *> &d_xs,
                           const std::vector<void *> &d_diff, size_t capacity) {
        check(mi355_group_diff_stream_batch(g_, d_frames.data(), stride, nframes, d_offsets.data(), d_xs.data(),
                                            d_diff.data(), capacity), "mi355_group_diff_stream_batch");
    }
    // the final changed-pixel gather to device `root`; returns every device's count
    std::vector<uint64_t> gather(int root, int nframes, const std::vector<void *> &d_offsets,
                                 const std::vector<void *> &d_xs, const std::vector<void *> &d_diff,
                                 size_t member_capacity, void *d_root_offsets, void *d_root_xs, void *d_root_diff,
                                 size_t root_capacity) {
        std::vector<uint64_t> counts((size_t)size());
        std::vector<const void *> o(d_offsets.begin(), d_offsets.end()), x(d_xs.begin(), d_xs.end()),
            d(d_diff.begin(), d_diff.end());
        check(mi355_group_gather(g_, root, nframes, o.data(), x.data(), d.data(), member_capacity, d_root_offsets,
                                 d_root_xs, d_root_diff, root_capacity, counts.data()), "mi355_group_gather");
        return counts;
    }
    void synchronize() { check(mi355_group_synchronize(g_), "mi355_group_synchronize"); }
};

}  // namespace cuda
}  // namespace diff
#endif
