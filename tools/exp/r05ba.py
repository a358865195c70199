# r05ba: how long does the GPU sit idle between bench.py's warm-up steps and its timed steps (host calls in between)?
import time, torch
from cudavideostream_amd.core import CUDACore
from cudavideostream_amd import synth
W, H, B = 1920, 1080, 256
n = 3 * W * H
dev = torch.device("cuda:0")
base, frames = synth.webcam_stream(B, W, H, seed=21, device=dev)
cap = max(B * n // 8, 1 << 20)
d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
with CUDACore(W, H, max_batch=B) as core:
    core.set_state(base.cpu().numpy())
    torch.cuda.synchronize()
    for trial in range(4):
        time.sleep(0.3)
        for i in range(5):
            core.diff_stream_batch(frames, B, d_off, d_xs, d_df, cap)
        t_a = time.perf_counter()
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        core.set_timing(True)
        core.reset_timing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            core.diff_stream_batch(frames, B, d_off, d_xs, d_df, cap)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        core.set_timing(False)
        print("trial %d: warm-up drained in %.2f ms, host gap before the timed steps %.3f ms, 20 steps %.4f ms each" % (trial, (t_b - t_a) * 1e3, (t0 - t_b) * 1e3, (t1 - t0) / 20 * 1e3), flush=True)
