// tools/group_demo.cpp -- a C++ server's view of several GPUs, in plain C++ over the C-ABI only (g++, no HIP
// header): mi355_group_create over ndev devices, every device runs its own stateful stream,
// one mi355_group_gather brings the changed-pixel streams to device 0 over RCCL/xGMI.
//
// Checks (exit status 0 = all passed): the root's copy of every rank's index, indices and differences equals
// what that rank produced, the counts add up, and the root's device holds them back to back in rank order.
// With one GPU (ndev 1) the same calls run and the gather degenerates to the root's own copy.
//
//   tools/group_demo [--ndev D] [--root R] [--width W] [--height H] [--frames T] [--same-device 1]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/mi355diff.h"

#define OK(call)                                                                              \
    do {                                                                                      \
        if ((call) != MI355_OK) {                                                             \
            fprintf(stderr, "%s failed: %s\n", #call, mi355_last_error());                    \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

static uint32_t rng_state = 777;
static uint32_t rnd() { return rng_state = rng_state * 1664525u + 1013904223u; }

int main(int argc, char **argv) {
    int ndev = 1, W = 320, H = 180, T = 6, root = 0;
    bool same_device = false;   // every rank on device 0: only with a stand-in for RCCL (MI355_RCCL_LIB), see tests/mock_rccl
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--ndev")) ndev = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--root")) root = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--same-device")) same_device = atoi(argv[i + 1]) != 0;
        else if (!strcmp(argv[i], "--width")) W = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--height")) H = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--frames")) T = atoi(argv[i + 1]);
    }
    const size_t n = (size_t)3 * W * H, cap = n * T;
    mi355_config cfg{};
    cfg.width = W; cfg.height = H; cfg.threshold = 20; cfg.max_batch = T; cfg.device = -1;
    mi355_group *grp = nullptr;
    std::vector<int> devs(ndev, 0);
    OK(mi355_group_create(&cfg, ndev, same_device ? devs.data() : nullptr, &grp));
    if (mi355_group_ranks(grp) != ndev || mi355_group_local_members(grp) != ndev) return 2;

    std::vector<void *> d_frames(ndev), d_off(ndev), d_xs(ndev), d_df(ndev);
    std::vector<std::vector<uint8_t>> frames(ndev);
    for (int r = 0; r < ndev; r++) {
        mi355_core *c = mi355_group_core(grp, r);
        if (mi355_group_rank_of(grp, r) != r || !c) return 2;
        // every rank its own stream: noise below the threshold plus a block that moves, different per rank
        std::vector<uint8_t> base(n);
        for (auto &b : base) b = (uint8_t)(40 + (rnd() >> 24) % 100);
        frames[r].resize(n * T);
        for (int t = 0; t < T; t++) {
            uint8_t *f = frames[r].data() + n * t;
            for (size_t i = 0; i < n; i++) f[i] = (uint8_t)(base[i] + (rnd() >> 24) % 9);
            const int bw = W / 5 + 1 + r, bh = H / 5 + 1, x0 = (t * 5 + 7 * r) % (W - bw), y0 = H / 4;
            for (int y = y0; y < y0 + bh; y++)
                for (int x = x0; x < x0 + bw; x++)
                    for (int ch = 0; ch < 3; ch++) f[((size_t)y * W + x) * 3 + ch] = (uint8_t)(210 + 10 * ch);
        }
        OK(mi355_set_state(c, base.data()));
        OK(mi355_dev_alloc(c, &d_frames[r], n * T));
        OK(mi355_dev_alloc(c, &d_off[r], sizeof(uint32_t) * (T + 1)));
        OK(mi355_dev_alloc(c, &d_xs[r], sizeof(int32_t) * cap));
        OK(mi355_dev_alloc(c, &d_df[r], cap));
        OK(mi355_upload(c, d_frames[r], frames[r].data(), n * T));
    }
    mi355_core *root_core = mi355_group_core(grp, root);
    void *r_off = nullptr, *r_xs = nullptr, *r_df = nullptr;
    const size_t rcap = cap * ndev;
    OK(mi355_dev_alloc(root_core, &r_off, sizeof(uint32_t) * (T + 1) * ndev));
    OK(mi355_dev_alloc(root_core, &r_xs, sizeof(int32_t) * rcap));
    OK(mi355_dev_alloc(root_core, &r_df, rcap));

    OK(mi355_group_diff_stream_batch(grp, d_frames.data(), n, T, d_off.data(), d_xs.data(), d_df.data(), cap));
    std::vector<uint64_t> counts(ndev);
    OK(mi355_group_gather(grp, root, T, d_off.data(), d_xs.data(), d_df.data(), cap, r_off, r_xs, r_df, rcap, counts.data()));
    OK(mi355_group_synchronize(grp));

    // what the root received against what every rank holds
    std::vector<uint32_t> all_off((size_t)(T + 1) * ndev);
    OK(mi355_download(root_core, all_off.data(), r_off, all_off.size() * sizeof(uint32_t)));
    size_t at = 0;
    uint64_t total = 0;
    for (int r = 0; r < ndev; r++) {
        mi355_core *c = mi355_group_core(grp, r);
        std::vector<uint32_t> off(T + 1);
        OK(mi355_download(c, off.data(), d_off[r], off.size() * sizeof(uint32_t)));
        if (off[T] != counts[r] || off[0] != 0) { fprintf(stderr, "rank %d: count mismatch\n", r); return 3; }
        if (memcmp(off.data(), all_off.data() + (size_t)r * (T + 1), off.size() * sizeof(uint32_t))) { fprintf(stderr, "rank %d: index differs at the root\n", r); return 3; }
        const size_t cnt = off[T];
        std::vector<int32_t> xs(cnt), gxs(cnt);
        std::vector<uint8_t> df(cnt), gdf(cnt);
        OK(mi355_download(c, xs.data(), d_xs[r], cnt * sizeof(int32_t)));
        OK(mi355_download(c, df.data(), d_df[r], cnt));
        OK(mi355_download(root_core, gxs.data(), (const int32_t *)r_xs + at, cnt * sizeof(int32_t)));
        OK(mi355_download(root_core, gdf.data(), (const uint8_t *)r_df + at, cnt));
        if (xs != gxs || df != gdf) { fprintf(stderr, "rank %d: payload differs at the root\n", r); return 3; }
        if (cnt == 0) { fprintf(stderr, "rank %d: empty stream, nothing was checked\n", r); return 3; }
        at += cnt;
        total += cnt;
    }
    printf("{\"tool\": \"group_demo\", \"ndev\": %d, \"root\": %d, \"width\": %d, \"height\": %d, \"frames\": %d, \"entries_at_root\": %llu, \"ok\": true}\n",
           ndev, root, W, H, T, (unsigned long long)total);
    for (int r = 0; r < ndev; r++) {
        mi355_core *c = mi355_group_core(grp, r);
        OK(mi355_dev_free(c, d_frames[r])); OK(mi355_dev_free(c, d_off[r])); OK(mi355_dev_free(c, d_xs[r])); OK(mi355_dev_free(c, d_df[r]));
    }
    OK(mi355_dev_free(root_core, r_off)); OK(mi355_dev_free(root_core, r_xs)); OK(mi355_dev_free(root_core, r_df));
    mi355_group_destroy(grp);
    return 0;
}
