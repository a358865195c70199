import sys, time, torch
sys.path.insert(0, '.')
from cudavideostream_amd import CUDACore, lib
W, H, B = 1920, 1080, 96
n = 3 * W * H
dev = torch.device('cuda', 0)
out = torch.empty((B, n), dtype=torch.uint8, device=dev)
core = CUDACore(W, H, max_batch=B)
core.use_torch_stream()
for name, fr in (("flat 128", torch.full((B, n), 128, dtype=torch.uint8, device=dev)),
                 ("two-level", (torch.arange(n, device=dev) // 3 % 2 * 200).to(torch.uint8).repeat(B, 1)),
                 ("random", torch.randint(0, 256, (B, n), dtype=torch.uint8, device=dev))):
    for _ in range(3):
        core.filter_batch(lib.OP_GRAY_WEIGHTED_BINARIZE, fr, out, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        core.filter_batch(lib.OP_GRAY_WEIGHTED_BINARIZE, fr, out, B)
    torch.cuda.synchronize()
    print(name, "fused gray+binarize %.3f us per frame" % ((time.perf_counter() - t0) * 1e6 / (10 * B)))
