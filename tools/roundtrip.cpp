// tools/roundtrip.cpp -- server -> socket bytes -> client, in plain C++ over the C-ABI only.
//
// Compiled with g++ (no HIP header, no HIP runtime call here): what a host written in any language with
// a C FFI does.  A "server" core packs a stream of frames into the byte stream the reference's sender
// thread writes (server/src/threads.cpp:220-233: base frame, then per frame u32 n | i32 xs[n] | u8 diff[n]);
// the bytes go through a pipe to a "client" that parses them exactly as client/opencv.cpp:38-66 does
// (read the base frame, then pos, pos indices, pos differences per frame) and rebuilds the frames on a
// second core.  The program checks that the client's frame equals the server's reconstructed state
// after every batch and that every rebuilt byte is within the threshold of the frame that was sent.
//
//   tools/roundtrip [--width W] [--height H] [--frames T] [--batch B]     exit status 0 = all checks passed
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include "../include/mi355diff.h"

#define OK(call)                                                                              \
    do {                                                                                      \
        if ((call) != MI355_OK) {                                                             \
            fprintf(stderr, "%s failed: %s\n", #call, mi355_last_error());                    \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

static uint32_t rng_state = 12345;
static uint32_t rnd() { return rng_state = rng_state * 1664525u + 1013904223u; }

// a frame sequence with sub-threshold noise everywhere and a block that moves
static void make_frame(std::vector<uint8_t> &f, const std::vector<uint8_t> &base, int w, int h, int t) {
    for (size_t i = 0; i < f.size(); i++) f[i] = (uint8_t)(base[i] + (rnd() >> 24) % 9);
    const int bw = w / 4 + 1, bh = h / 4 + 1, x0 = (t * 3) % (w - bw + 1), y0 = h / 3;
    for (int y = y0; y < y0 + bh && y < h; y++)
        for (int x = x0; x < x0 + bw; x++)
            for (int c = 0; c < 3; c++) f[((size_t)y * w + x) * 3 + c] = (uint8_t)(200 + 10 * c);
}

static bool read_all(int fd, void *p, size_t n) {   // the client's read loops, opencv.cpp:40-62
    uint8_t *b = (uint8_t *)p;
    while (n) {
        const ssize_t k = read(fd, b, n);
        if (k <= 0) return false;
        b += k;
        n -= (size_t)k;
    }
    return true;
}

// "send" n bytes and "receive" them: written in pieces that fit the pipe buffer, read back in between
static bool through_pipe(int wfd, int rfd, const uint8_t *src, uint8_t *dst, size_t n) {
    for (size_t at = 0; at < n;) {
        const size_t piece = n - at < 32768 ? n - at : 32768;
        if (write(wfd, src + at, piece) != (ssize_t)piece) return false;
        if (!read_all(rfd, dst + at, piece)) return false;
        at += piece;
    }
    return true;
}

int main(int argc, char **argv) {
    int w = 320, h = 180, T = 24, B = 8;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const int v = atoi(argv[i + 1]);
        if (k == "--width") w = v; else if (k == "--height") h = v;
        else if (k == "--frames") T = v; else if (k == "--batch") B = v;
    }
    const size_t n = (size_t)3 * w * h;
    mi355_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.width = w; cfg.height = h; cfg.threshold = 20; cfg.max_batch = B; cfg.device = -1;
    mi355_core *server = nullptr, *client = nullptr;
    OK(mi355_create(&cfg, &server));
    OK(mi355_create(&cfg, &client));

    std::vector<uint8_t> base(n), frame(n), frames((size_t)B * n), shown((size_t)B * n), s_state(n), c_state(n);
    for (size_t i = 0; i < n; i++) base[i] = (uint8_t)(40 + (i * 7) % 150);
    OK(mi355_set_state(server, base.data()));                  // kernels.cu:406

    int fds[2];
    if (pipe(fds) != 0) return 1;
    // ---- sender side: the base frame first (threads.cpp:220)
    std::vector<uint8_t> wire_host(mi355_wire_bytes(B, (uint64_t)B * n));
    void *d_frames = nullptr, *d_wire = nullptr, *d_off = nullptr, *d_cwire = nullptr, *d_shown = nullptr;
    OK(mi355_dev_alloc(server, &d_frames, (size_t)B * n));
    OK(mi355_dev_alloc(server, &d_wire, wire_host.size()));
    OK(mi355_dev_alloc(server, &d_off, sizeof(uint32_t) * (B + 1)));
    OK(mi355_dev_alloc(client, &d_cwire, wire_host.size()));
    OK(mi355_dev_alloc(client, &d_shown, (size_t)B * n));

    // ---- client start-up: read the base frame (opencv.cpp:38-46)
    std::vector<uint8_t> got_base(n);
    if (!through_pipe(fds[1], fds[0], base.data(), got_base.data(), n)) return 1;
    OK(mi355_set_state(client, got_base.data()));

    size_t sent_bytes = 0, changed = 0;
    int max_err = 0;
    for (int t0 = 0; t0 < T; t0 += B) {
        const int nb = T - t0 < B ? T - t0 : B;
        for (int k = 0; k < nb; k++) {
            make_frame(frame, base, w, h, t0 + k);
            memcpy(&frames[(size_t)k * n], frame.data(), n);
        }
        // ---- server: one batch -> the socket bytes of nb frames
        std::vector<uint32_t> off(nb + 1);
        OK(mi355_upload(server, d_frames, frames.data(), (size_t)nb * n));
        OK(mi355_diff_stream_wire_batch(server, d_frames, n, nb, d_off, d_wire, wire_host.size()));
        OK(mi355_download(server, off.data(), d_off, sizeof(uint32_t) * (nb + 1)));
        const size_t wb = mi355_wire_bytes(nb, off[nb]);
        OK(mi355_download(server, wire_host.data(), d_wire, wb));
        changed += off[nb];
        // ---- the socket (a pipe here)
        std::vector<uint8_t> rx(wb);
        if (!through_pipe(fds[1], fds[0], wire_host.data(), rx.data(), wb)) return 1;
        sent_bytes += wb;
        // ---- client: parse the headers as opencv.cpp:52 does, hand the bytes to the device
        std::vector<uint32_t> counts(nb);
        size_t at = 0;
        for (int k = 0; k < nb; k++) {
            uint32_t pos;
            memcpy(&pos, &rx[at], 4);
            counts[k] = pos;
            at += 4 + (size_t)5 * pos;
        }
        if (at != wb) { fprintf(stderr, "stream framing broken\n"); return 1; }
        OK(mi355_upload(client, d_cwire, rx.data(), wb));
        OK(mi355_apply_wire_batch(client, d_cwire, counts.data(), nb, d_shown, n));
        OK(mi355_download(client, shown.data(), d_shown, (size_t)nb * n));
        // ---- checks
        OK(mi355_get_state(server, s_state.data()));
        OK(mi355_get_state(client, c_state.data()));
        if (memcmp(s_state.data(), c_state.data(), n) != 0) { fprintf(stderr, "client state != server state\n"); return 1; }
        if (memcmp(&shown[(size_t)(nb - 1) * n], c_state.data(), n) != 0) { fprintf(stderr, "last shown frame != state\n"); return 1; }
        for (size_t i = 0; i < (size_t)nb * n; i++) {
            const int e = abs((int)shown[i] - (int)frames[i]);
            if (e > max_err) max_err = e;
        }
    }
    if (max_err > cfg.threshold) { fprintf(stderr, "rebuilt frame off by %d > threshold\n", max_err); return 1; }
    OK(mi355_dev_free(server, d_frames));
    OK(mi355_dev_free(server, d_wire));
    OK(mi355_dev_free(server, d_off));
    OK(mi355_dev_free(client, d_cwire));
    OK(mi355_dev_free(client, d_shown));
    mi355_destroy(server);
    mi355_destroy(client);
    printf("{\"roundtrip\": \"ok\", \"width\": %d, \"height\": %d, \"frames\": %d, \"batch\": %d, "
           "\"changed_bytes\": %zu, \"wire_bytes\": %zu, \"raw_bytes\": %zu, \"max_abs_error\": %d}\n",
           w, h, T, B, changed, sent_bytes, (size_t)T * n, max_err);
    return 0;
}
