#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric : 1920x1080 RGB24 frames/sec through diff + threshold + pack (+ achieved HBM GB/s vs peak)
config : configs[1] "1080p30 synthetic stream, diff+threshold+pack on 1xMI355X": the S1 `webcam`
         stream resident in HBM as a batch of B frames; one step = one pass of the
         hot path (mi355_diff_stream_batch: pack kernel + scans + gather kernel) over that batch,
         the stream state carried from step to step.  Inputs are already in HBM when the timed region
         starts.
N GPUs : one process per GPU (torch.distributed, backend nccl = RCCL); every rank runs an independent
         stream of the same shape (weak scaling, no data-path collective): the timed region is EXACTLY K steps of
         the hot path between barriers, MAX over the ranks.  The path's one exchange step -- the final changed-pixel
         gather: ONE gather-v of the final batch's payload to rank 0 below the C-ABI (mi355_group_gather) -- runs
         right behind the timed steps, timed on its own (--gather after, the default: final_gather_ms,
         value_with_final_gather); --gather last puts it inside the timed region, every / index / none as named.

         `python bench.py --gpus N` starts the N ranks itself when no launcher has (WORLD_SIZE unset); under
         `python -m torch.distributed.run ... bench.py --gpus N` this process is one of them.  With a group the line also
         carries ranks_seen, gather_ms / gather_bytes, gather_verified and `gather_every` (the same job with a gather
         after EVERY batch).  If the RCCL group below the C-ABI cannot be formed, a run whose exchange lies INSIDE the
         timed steps (--gather last / every) FAILS (exit 3) unless --allow-gather-fallback is given; with the default
         (--gather after: the exchange behind the timed steps) the steps are measured all the same, the exchange runs in
         its torch.distributed form, and config.gather_impl says so with the library's error.
config 5 (BASELINE.json configs[4], 4K frames dealt round-robin over the GPUs, RCCL gather over xGMI): whenever a launcher
         started the job (N > 1; N = 1 under torch.distributed.run) the line carries a `config5` object measured ACROSS the
         ranks -- a 3840x2160 sequence dealt round-robin over the ranks that are there, 64-frame shards, stateless pairs,
         then mi355_group_gather to rank 0: frames/s (MAX over the ranks), frac per rank, final_gather_ms, gather_bytes,
         gather_gbps, ranks_seen, gather_verified, and parity of rank 0's copy of every rank's first and last frame against
         the oracle.  At N = 1 without a launcher: `config5_per_gpu` (one GPU's share of the same deal over 8 ranks).  The
         same shape as the WHOLE job: python bench.py --gpus 8 --shard roundrobin --width 3840 --height 2160 --batch 64
timing : `value` = the W warm-up + K timed steps behind --preheat-s seconds (default 3) of the same step, untimed: a chip that
         has been idle needs ~12 ms of load to reach its clocks and a second or two to settle; the same W + K steps as the first GPU
         work of the process are reported beside it (`cold_start_window`: what rounds 1-5 printed), and `steady_state` (1000
         more steps) behind it.  roofline.frac is on the ALGORITHMIC bytes (2N + 5P); frac_actual / frac_of_achievable on the
         bytes the chip moved (counters) over 8 TB/s / over this board's plain streaming read.

Prints ONE JSON line on rank 0.  The CPU oracle is used only as the checker / cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

# Under a launcher this process holds torch.distributed's and RCCL's stream pools AND two cores (the job's and config5's), each
# with three streams that must run beside each other: the HIP runtime serves the streams of one priority class with at most
# GPU_MAX_HW_QUEUES hardware queues (default 4) -- raised BEFORE the runtime starts, and the cores are created with
# MI355_FLAG_OWN_QUEUES (include/mi355diff.h).  Measured with one rank under the launcher: headline 462 k frames/s without
# either, 583 k with the flag; config5 0.52 of the roofline with the flag alone, 0.62 with both (profiles/README.md, r06k-r06p).
if "RANK" in os.environ:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np   # noqa: E402
import torch   # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cudavideostream_amd import CUDACore, synth  # noqa: E402
from cudavideostream_amd import gather as gx  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
KERNELS = ("k_diff_pack", "k_scan_groups", "k_expand")   # the path's three kernels, in launch order


def lib_sha256():
    import hashlib
    from cudavideostream_amd import lib as _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    except Exception:
        return None


def pmc_counters(B, W, H):
    """Counter bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_summary.json), or None
    when they were taken at another configuration or on another build of the library (stale numbers are not
    reported)."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None
    if (rec.get("batch"), rec.get("width"), rec.get("height")) != (B, W, H):
        return None
    # the counters belong to one build of the library: the file's hash, or (the file differs between build directories)
    # the hash of the sources it was built from (profiles/srchash.py)
    same_lib = rec.get("lib_sha256") and rec["lib_sha256"] == lib_sha256()
    same_src = False
    if not same_lib and rec.get("src_sha256"):
        try:
            sys.path.insert(0, os.path.join(ROOT, "profiles"))
            from srchash import src_sha256
            same_src = rec["src_sha256"] == src_sha256(ROOT)
        except Exception:   # noqa: BLE001
            same_src = False
    if not (same_lib or same_src):
        return None
    return rec


def path_roofline(alg_bytes, ms, launches, pair, pmc=None, wall_ms=None):
    """roofline object of one configuration of the path.  achieved = ALGORITHMIC bytes (2N + 5P per
    frame) over the time of ALL kernels of the path (pack + scan + expand, HIP events on the streams they run on).
    wall_ms: pipelined batches (the expansion of batch k runs beside the pack kernel of batch k + 1, so the three
    durations overlap and stretch each other): the denominator is then the time one batch takes in the steady
    state -- the timed region's wall clock per step, the larger and the honest one."""
    per = [m / max(launches, 1) for m in ms]
    total_ms = wall_ms if wall_ms is not None else sum(per)
    achieved = alg_bytes / (total_ms * 1e-3) / 1e9
    # template arguments: PAIR, ALIGNED, HIGH (threshold >= 128), ONCE (non-temporal frame loads: the frames of a stream,
    # pairs whose operands share no frame -- as the two halves / separate buffers the bench hands in)
    names = ("mi355::k_diff_pack<%s,true,false,true>" % ("true" if pair else "false"), "mi355::k_scan_groups",
             "mi355::k_expand<false>")
    kernels = []
    for short, name, m in zip(KERNELS, names, per):
        k = {"kernel": name, "avg_us": round(m * 1e3, 2)}
        c = pmc["kernels"].get(short) if pmc else None
        if c:
            moved = c["read_bytes"] + c["write_bytes"]
            k.update({"bytes_moved": moved, "read_bytes": c["read_bytes"], "write_bytes": c["write_bytes"],
                      "frac_of_peak": round(moved / (m * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
        kernels.append(k)
    frac = achieved / HBM_PEAK_GBPS
    assert frac > 0.0, f"roofline fraction {frac}"
    out = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(frac, 4),
           "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
           "kernel": "+".join(KERNELS), "kernel_ms": round(total_ms, 4),
           "algorithmic_bytes_per_launch": int(alg_bytes), "kernels": kernels}
    if wall_ms is not None:
        out["kernel_ms_basis"] = ("wall clock per step of the timed region: consecutive batches are pipelined (index + "
                                  "expansion of batch k on a side stream beside the pack kernel of batch k + 1), the "
                                  f"kernels' own durations overlap and add up to {round(sum(per), 4)} ms")
    if pmc:
        out["traffic_source"] = f"profiles/pmc_summary.json ({pmc.get('tag')}, separate --pmc FETCH_SIZE / WRITE_SIZE passes, same library build)"
    out.update(fractions(alg_bytes, pmc["hbm_bytes_per_launch"] if pmc else None, total_ms * 1e-3, pair))
    return out


def fractions(alg_bytes, traffic_bytes, sec_per_launch, pair, stream_gbps=None):
    """The two truths of a roofline line, side by side.  `frac` (set by the caller) is the contract's: ALGORITHMIC bytes
    (2N + 5P per frame) / time / 8 TB/s.  It is not HBM utilisation: in stream mode the pack kernel keeps
    the state in registers across the batch and reads N per frame where the model counts 2N, so the chip moves fewer
    bytes than the model -- and `frac` can pass 1 on a cool chip without anything being skipped.  `frac_actual` is the
    bytes the chip really moved (FETCH_SIZE / WRITE_SIZE counters) / time / 8 TB/s, `frac_of_achievable` the same over
    what a plain streaming read reaches on THIS board (board.hbm_stream_read_gbps, measured in the same run)."""
    out = {}
    if traffic_bytes:
        gbps = traffic_bytes / sec_per_launch / 1e9
        out["actual_gbps"] = round(gbps, 1)
        out["frac_actual"] = round(gbps / HBM_PEAK_GBPS, 4)
        out["traffic_over_algorithmic"] = round(traffic_bytes / alg_bytes, 4)
        if stream_gbps:
            out["frac_of_achievable"] = round(gbps / stream_gbps, 4)
    out["frac_note"] = ("frac = algorithmic 2N + 5P bytes / time / 8 TB/s"
                        + ("" if pair else "; the stream kernel keeps the state in registers and reads N per frame, not 2N: the chip "
                           "moves traffic_over_algorithmic of those bytes, so frac is NOT HBM utilisation and can pass 1")
                        + "; frac_actual = counter bytes / time / 8 TB/s; frac_of_achievable = the same / this board's plain "
                          "streaming read (board.hbm_stream_read_gbps)")
    return out


def smi_snapshot():
    """sclk / mclk / power of the board as rocm-smi reports them (outside every timed region); None where the tool is
    missing or refuses.  Explains nothing by itself under load (the guide: in-kernel clocks read up to 10 % below
    pp_dpm_sclk), so the in-kernel probes of `board` go with it."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
        d = json.loads(r.stdout.decode(errors="replace") or "{}")
        card = d.get("card0") or (next(iter(d.values())) if d else {})
        keep = {}
        for k, v in card.items():
            lk = k.lower()
            if any(t in lk for t in ("sclk", "mclk", "fclk", "power", "performance level")) and "socclk" not in lk:
                keep[k] = v
        return keep or None
    except Exception:   # noqa: BLE001
        return None


def board_fingerprint(core, when):
    """What this board holds under load, measured inside kernels right after the timed region (the chip is warm):
    the shader clock under an integer-VALU load and the GB/s of a plain streaming read.  The path's frames/s follow the
    second (the pack kernel is bound by the memory system), and boards of the pool differ by ~7 % in it."""
    out = {"when": when}
    try:
        out["shader_mhz_under_valu_load"] = round(core.probe_clock(100), 0)
        out["hbm_stream_read_gbps"] = round(core.probe_hbm_read(2048), 0)
        out["hbm_stream_write_gbps"] = round(core.probe_hbm_write(2048), 0)
        out["hbm_narrow_write_gbps"] = round(core.probe_hbm_write(2048, narrow=True), 0)   # index + value per lane: the dense expansion's stores
    except Exception as e:   # noqa: BLE001
        out["skipped"] = repr(e)[:120]
    return out


def metric_name(W, H):
    """BASELINE.json's metric, verbatim for its 1080p configuration (same wording at other sizes)."""
    name = "1920\u00d71080 RGB24 frames/sec (diff+threshold+pack); achieved HBM GB/s vs peak"
    try:
        name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        pass
    return name if (W, H) == (1920, 1080) else name.replace("1920\u00d71080", f"{W}\u00d7{H}")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=256, help="frames per step (resident in HBM)")
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--gather", choices=["after", "last", "every", "index", "none"], default="after",
                   help="the job's exchange step (N > 1, or N = 1 under a launcher): gather-v of a batch's changed-pixel stream "
                        "to rank 0 below the C-ABI.  after (default): ONE gather of the final batch right behind the K timed "
                        "steps, timed on its own (gather_ms, gather_bytes, gather_verified; value_with_final_gather puts it "
                        "into the denominator); last: the same gather INSIDE the timed region (rounds 1-4); every: after "
                        "every batch; index: the per-frame index every step, the payload with the last; none")
    p.add_argument("--shard", choices=["streams", "roundrobin"], default="streams",
                   help="streams: every rank owns an independent stateful stream (the headline workload); "
                        "roundrobin: the frames of ONE sequence are dealt to the ranks and diffed against "
                        "their raw predecessors, stateless (BASELINE config 5)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-pair", action="store_true")
    p.add_argument("--no-host-path", action="store_true")
    p.add_argument("--no-filters", action="store_true", help="skip the config3 / config4 objects")
    p.add_argument("--allow-gather-fallback", action="store_true",
                   help="N > 1 only: if the RCCL group below the C-ABI cannot be formed, measure the torch.distributed "
                        "form of the exchange instead of failing (the line then says so in config.gather_impl)")
    p.add_argument("--steady-steps", type=int, default=1000,
                   help="the step repeated this many times behind the timed region, no exchange (`steady_state`; N > 1: every "
                        "rank, MAX over the ranks, frames_per_s of the whole job); 0: skip")
    p.add_argument("--gather-every-steps", type=int, default=10,
                   help="N > 1: steps of the secondary `gather_every` measurement (a gather after EVERY batch)")
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="NOT a measurement: run the N-rank job with every rank on device 0 -- one process per rank, "
                        "torch.distributed over gloo, the exchange below the C-ABI through the tests' inter-process "
                        "stand-in for RCCL (tests/mock_rccl/librccl_mock_ipc.so; real RCCL refuses two ranks on one "
                        "device) -- to prove the process-per-rank sequence before the first run on an 8-GPU node")
    p.add_argument("--no-config5", action="store_true",
                   help="skip BASELINE configs[4]: the config5_per_gpu object (one GPU's 4K round-robin shard, N = 1) and the "
                        "config5 object (the 4K sequence dealt over the ranks of this job + the gather, under a launcher)")
    p.add_argument("--config5-size", type=int, nargs=3, default=[3840, 2160, 64], metavar=("W", "H", "FRAMES"),
                   help="config5: frame size and frames per rank's shard (the tests take a small one)")
    p.add_argument("--config5-steps", type=int, default=20, help="config5: timed passes over the shard")
    p.add_argument("--own-queues", choices=["auto", "0", "1"], default="auto",
                   help="MI355_FLAG_OWN_QUEUES on the cores: auto (default) = under a launcher only (a process that also holds "
                        "torch.distributed's / RCCL's stream pools), 0 / 1 = never / always")
    p.add_argument("--preheat-s", type=float, default=3.0,
                   help="seconds of the same step, untimed, in front of the W warm-up + K timed steps (the chip's clock ramp and "
                        "first heating: DESIGN.md section 8); 0: none -- the W + K steps are then the first GPU work of the process, "
                        "as in rounds 1-5 (the default run reports that window too: cold_start_window)")
    return p.parse_args()


def cpu_baseline(args, base, frames, dev):
    """Oracle (CPU restatement of tests/cuda_streaming/test.cu:560-576) timed on a bounded sample of
    the same workload, single-threaded like the reference's one elaboration thread (the frames of the
    batch, cycled with the state carried along, for about --cpu-seconds); also the parity check of the
    GPU path on the first frames of the stream."""
    from oracle import pyoracle as po
    B = frames.shape[0]
    h_base = base.cpu().numpy()
    h_frames = frames.cpu().numpy()
    S = min(B, 64)
    eo, exs, edf, est = po.diff_stream(h_frames[:S], h_base)
    # parity of the product path on the same sample
    with CUDACore(args.width, args.height, max_batch=S) as c2:
        c2.use_torch_stream()
        c2.set_state(h_base)
        cap = int(eo[-1]) + 16
        d_off = torch.zeros(S + 1, dtype=torch.int32, device=dev)
        d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
        d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
        c2.diff_stream_batch(frames, S, d_off, d_xs, d_df, cap)
        torch.cuda.synchronize()
        ok = (np.array_equal(d_off.cpu().numpy().view(np.uint32), eo)
              and np.array_equal(d_xs[:int(eo[-1])].cpu().numpy(), exs)
              and np.array_equal(d_df[:int(eo[-1])].cpu().numpy(), edf)
              and np.array_equal(c2.get_state(), est))
    # timing: single thread
    L = po.lib()
    n = h_base.size
    st = h_base.copy()
    xs = np.empty(n, np.int32)
    df = np.empty(n, np.uint8)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.cpu_seconds:
        for t in range(B):
            L.ora_diff_pack(h_frames[t], st, n, 20, xs, df)
        done += B
    dt = time.perf_counter() - t0
    # timing: ALL host cores -- a row band per thread for all frames of a sub-batch (a band's state belongs to its thread:
    # the threads are started once per sub-batch, not per frame), pieces put together in (frame, band) order: output
    # identical to the single-threaded loop (tests/test_oracle.py::test_stream_mt_matches_single_thread)
    ncores = os.cpu_count() or 1
    nthr = max(1, min(ncores, 1024))
    SB = min(B, 64)
    st = h_base.copy()
    cap_mt = SB * n
    xs_mt = np.empty(cap_mt, np.int32)
    df_mt = np.empty(cap_mt, np.uint8)
    off_mt = np.zeros(SB + 1, np.uint32)
    sub = np.ascontiguousarray(h_frames[:SB]).reshape(-1)
    done_mt, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < min(4.0, args.cpu_seconds):
        rc = L.ora_diff_stream_mt(sub, SB, st, n, 20, off_mt, xs_mt, df_mt, cap_mt, nthr)
        assert rc == 0, rc
        done_mt += SB
    dt_mt = time.perf_counter() - t0
    base_obj = {"value": round(done / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": f"{done} frames ({done // B} passes over the batch's {B} frames of the same "
                          f"{args.width}x{args.height} S1 stream, state carried), oracle/cpu_ref.c "
                          f"ora_diff_pack, {dt:.1f} s",
                "all_cores": {"value": round(done_mt / dt_mt, 2), "unit": "frames/s", "cores": nthr,
                              "host_cores": ncores,
                              "sample": f"{done_mt} frames ({SB}-frame sub-batches, one row band per thread for the "
                                        f"whole sub-batch, {nthr} threads on {ncores} host cores), {dt_mt:.1f} s"}}
    ref = reference_filter_chain(args, h_base, h_frames)
    if ref:
        base_obj["reference_filter_chain"] = ref
    return base_obj, bool(ok)


def parity_of_secondary_lines(args, dev, frames):
    """One sample of every workload the line quotes a number for, product against oracle (bit-exact), outside the
    timed regions: pair mode, the three regimes, config 3, config 4.  (The headline's parity is cpu_baseline's.)"""
    from oracle import pyoracle as po
    from cudavideostream_amd import lib as L
    W, H = args.width, args.height
    n = 3 * W * H
    res = {}

    def pairs_ok(cur, prev):
        T = cur.shape[0]
        with CUDACore(W, H, max_batch=T) as c:
            c.use_torch_stream()
            cap = T * n
            d_off = torch.zeros(T + 1, dtype=torch.int32, device=dev)
            d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
            d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
            c.diff_pairs_batch(cur, prev, T, d_off, d_xs, d_df, cap)
            torch.cuda.synchronize()
            off = d_off.cpu().numpy().view(np.uint32)
            hx, hd = d_xs[:int(off[-1])].cpu().numpy(), d_df[:int(off[-1])].cpu().numpy()
        hc, hp = cur.cpu().numpy(), prev.cpu().numpy()
        ok = True
        for t in range(T):
            cnt, xs, df, _ = po.diff_pack(hc[t], hp[t])
            ok = ok and off[t + 1] - off[t] == cnt and np.array_equal(hx[off[t]:off[t + 1]], xs) and np.array_equal(hd[off[t]:off[t + 1]], df)
        return bool(ok)

    B = frames.shape[0] // 2
    res["pair"] = pairs_ok(frames[:2], frames[B:B + 2])
    rnd = torch.stack([synth.refrand_frame(n, 100 + t, device=dev) for t in (0, 1, 32, 33)])
    res["S0"] = pairs_ok(rnd[2:], rnd[:2])
    res["P_eq_N"] = pairs_ok(rnd[:2] ^ 0x80, rnd[:2])
    res["P_eq_0"] = pairs_ok(rnd[:2], rnd[:2].clone())

    # config 3 / config 4 on two frames of the filter workload's stream
    _, fr = synth.webcam_stream(3, W, H, seed=33, device=dev)
    h = fr.cpu().numpy()
    g = np.exp(-(np.arange(-1, 2)[:, None] ** 2 + np.arange(-1, 2)[None, :] ** 2) / (2.0 * 1.5 * 1.5))
    k = (g / g.sum()).astype(np.float32).reshape(-1)
    T = 2
    with CUDACore(W, H, k=k, max_batch=T) as c:
        c.use_torch_stream()
        cap = T * n
        d_off = torch.zeros(T + 1, dtype=torch.int32, device=dev)
        d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
        d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
        vis = torch.empty((T, n), dtype=torch.uint8, device=dev)
        filt = torch.empty((T, n), dtype=torch.uint8, device=dev)
        # config 3: weighted gray -> histogram -> two-max -> binarize, then the diff of the colour frames
        c.set_state(h[0])
        c.filter_batch(L.OP_GRAY_WEIGHTED_BINARIZE, fr[1:], vis, T)
        c.diff_stream_batch(fr[1:], T, d_off, d_xs, d_df, cap)
        torch.cuda.synchronize()
        eo, exs, edf, _ = po.diff_stream(h[1:], h[0])
        ok3 = np.array_equal(d_off.cpu().numpy().view(np.uint32), eo) and np.array_equal(d_xs[:int(eo[-1])].cpu().numpy(), exs) \
            and np.array_equal(d_df[:int(eo[-1])].cpu().numpy(), edf)
        for t in range(T):
            g3 = po.gray_weighted(h[1 + t])
            ok3 = ok3 and np.array_equal(vis[t].cpu().numpy(), po.binarize(g3, po.two_max_threshold(po.histogram(g3))))
        res["config3"] = bool(ok3)
        # config 4: noise filter, diff of the filtered frames, cleared red map from the packed indices
        c.set_state(h[0])
        c.filter_batch(L.OP_CONV3X3, fr[1:], filt, T)
        c.diff_stream_batch(filt, T, d_off, d_xs, d_df, cap)
        c.red_stream_batch(d_off, d_xs, T, vis)
        torch.cuda.synchronize()
        hf = np.stack([po.conv3x3(h[1 + t], W, H, k) for t in range(T)])
        eo, exs, edf, _ = po.diff_stream(hf, h[0])
        ok4 = np.array_equal(filt.cpu().numpy(), hf) and np.array_equal(d_off.cpu().numpy().view(np.uint32), eo) \
            and np.array_equal(d_xs[:int(eo[-1])].cpu().numpy(), exs) and np.array_equal(d_df[:int(eo[-1])].cpu().numpy(), edf)
        for t in range(T):
            ok4 = ok4 and np.array_equal(vis[t].cpu().numpy(), po.red_overlap(np.zeros(n, np.uint8), exs[eo[t]:eo[t + 1]]))
        res["config4"] = bool(ok4)
    # the 5x5 median (not on the server's path; its line stands beside the filter chains): a 640 x 360 crop of the same
    # stream through the same column-strip kernel (the oracle sorts 25 bytes per output: 10 s for a 1080p frame)
    cw, ch = 640, 360
    if W >= cw and H >= ch:
        crop = np.ascontiguousarray(h[1].reshape(H, W, 3)[:ch, :cw]).reshape(-1)
        with CUDACore(cw, ch) as c:
            c.use_torch_stream()
            d_o = torch.empty(crop.size, dtype=torch.uint8, device=dev)
            c.median5x5(torch.from_numpy(crop).to(dev), d_o)
            torch.cuda.synchronize()
            res["median5x5"] = bool(np.array_equal(d_o.cpu().numpy(), po.median5x5(crop, cw, ch)))
    return res


def reference_filter_chain(args, h_base, h_frames, nframes=8):
    """Part of the cpu_baseline leg: the reference's OWN CPU branch for the gray-avg -> histogram -> two-max ->
    binarize chain (server/src/server.cpp:96-135, compiled unmodified into oracle/_ref/server_cpu where the
    reference tree was present), timed by the reference's own per-frame `FOR:` counter (server.cpp:77,144,164)
    on this host.  The GPU counterpart is the fused chain of tools/bench_filters.py (3.7 us per 1080p frame)."""
    import re
    import subprocess
    import tempfile
    from oracle import pyoracle as po
    exe = po.ref_server_cpu_path()
    if exe is None:
        return None
    T = min(nframes, h_frames.shape[0])
    # the reference prints its counters once per second (server.cpp:151): loop the frames long enough
    repeat = min(4000, max(2, int(1500 * (1920 * 1080) / (args.width * args.height) / T)))   # ~1500 1080p frames
    try:
        with tempfile.TemporaryDirectory() as tmp:
            fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fin, "wb") as f:
                f.write(np.array([args.width, args.height, T], np.int32).tobytes())
                f.write(h_base.tobytes())
                f.write(h_frames[:T].tobytes())
            r = subprocess.run([exe], env=dict(os.environ, REF_IN=fin, REF_OUT=fout, REF_REPEAT=str(repeat)),
                               check=True, stdout=subprocess.PIPE, timeout=120)
    except Exception as e:  # a missing or foreign binary must not cost the bench line
        return {"skipped": repr(e)[:120]}
    ms = sorted(float(m) for m in re.findall(r"FOR:\s*([0-9.]+) ms", r.stdout.decode(errors="replace")))
    if not ms:
        return {"skipped": "no FOR: lines in the reference's output"}
    med = ms[len(ms) // 2]
    return {"value": round(1e3 / med, 1), "unit": "frames/s", "cores": 1, "kind": "reference",
            "ms_per_frame_median": round(med, 3),
            "sample": f"{T * repeat} frames through oracle/_ref/server_cpu (server.cpp CPU branch), "
                      f"{len(ms)} readings of its own FOR: counter"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1), BEFORE anything in this process touches the GPU, hand their one JSON line through
    and leave with their exit code.  (With WORLD_SIZE set -- the driver's `python -m torch.distributed.run ... bench.py`
    -- this process IS a rank and nothing is spawned.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    shm = None
    if args.rehearse_on_one_gpu:
        if args.gpus > 5:   # the GPU boxes of this project allow six processes on a card: five ranks + rank 0's rocm-smi
            raise SystemExit("bench.py --rehearse-on-one-gpu: at most 5 ranks on the one GPU")
        env.update(rehearsal_env())
        shm = env["MOCK_RCCL_SHM"]
    try:
        return subprocess.run(cmd, env=env).returncode
    finally:
        if shm:
            try:
                os.unlink("/dev/shm" + shm)
            except OSError:
                pass


def rehearsal_env():
    """Environment of a --rehearse-on-one-gpu job: the inter-process stand-in for RCCL and the shared-memory object its
    ranks meet in (an existing MOCK_RCCL_SHM -- a test's -- is kept)."""
    mock = os.path.join(ROOT, "tests", "mock_rccl", "librccl_mock_ipc.so")
    if not os.path.exists(mock):
        raise SystemExit(f"bench.py --rehearse-on-one-gpu: {mock} is missing (python -c 'import __graft_entry__ as g; g.build()')")
    return {"MI355_RCCL_LIB": os.environ.get("MI355_RCCL_LIB", mock),
            "MOCK_RCCL_SHM": os.environ.get("MOCK_RCCL_SHM", f"/mi355bench_{os.getpid()}"),
            # payloads are staged through shared host memory: room for a few ranks' batches at a time, and patience
            "MOCK_RCCL_SHM_MB": os.environ.get("MOCK_RCCL_SHM_MB", "256"),
            "MOCK_RCCL_TIMEOUT_S": os.environ.get("MOCK_RCCL_TIMEOUT_S", "60")}


def _device_sync():
    """(a no-op without a GPU: tests/test_gather_gloo.py runs the Exchange over gloo with host tensors)"""
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def payload_digest(o, xs, df):
    """[entries, sum of the offsets, weighted sums of the indices and of the differences] of one rank's packed batch."""
    pn = int(o[-1].item()) & 0xFFFFFFFF
    w = torch.arange(pn, device=xs.device, dtype=torch.int64) % 8191 + 1
    return [pn, int(o.to(torch.int64).sum().item()), int((xs[:pn].to(torch.int64) * w).sum().item()),
            int((df[:pn].to(torch.int64) * w).sum().item())]


class Exchange:
    """The job's ONE exchange step: gather-v of every rank's (offsets, xs, diff) of a batch to rank 0 -- below the C-ABI
    (mi355_group_gather: RCCL all-gather of counts + point-to-point sends to the root, csrc/group.hip) when the group
    could be formed, else its torch.distributed form (cudavideostream_amd/gather.py).  Every method is collective."""

    def __init__(self, group, dist, world, rank, B, cap, dev, cdev):
        self.group, self.dist, self.world, self.rank, self.B, self.cap, self.dev, self.cdev = group, dist, world, rank, B, cap, dev, cdev
        self.ms, self.calls, self.bytes, self.ranks_seen = 0.0, 0, 0, world
        self.root = None   # rank 0: (index[world, B + 1], xs of all ranks in rank order, diff likewise) of the latest gather
        if group is not None and rank == 0:
            self.r_off = torch.zeros((world, B + 1), dtype=torch.int32, device=dev)
            self.r_xs = torch.empty(world * cap, dtype=torch.int32, device=dev)
            self.r_df = torch.empty(world * cap, dtype=torch.uint8, device=dev)

    def reset(self):
        self.ms, self.calls, self.bytes = 0.0, 0, 0

    def run(self, d_off, d_xs, d_df):
        """One gather, timed on its own (a host synchronisation either side: the call has one inside anyway,
        kernels.cu:507-508)."""
        _device_sync()
        tg = time.perf_counter()
        if self.group is not None:
            root = self.rank == 0
            counts = self.group.gather(0, self.B, [d_off], [d_xs], [d_df], self.cap, self.r_off if root else None,
                                       self.r_xs if root else None, self.r_df if root else None, self.world * self.cap if root else 0)
            self.ranks_seen = self.group.nranks
            total = int(counts.sum())
            if root:
                self.root = (self.r_off, self.r_xs, self.r_df)
        else:
            totals, xs_all, df_all, index = gx.gather_payload(d_off, d_xs, d_df, dst=0)
            self.ranks_seen = len(totals)
            total = int(sum(totals))
            if self.rank == 0:
                self.root = (index, xs_all, df_all)
        _device_sync()
        ms = (time.perf_counter() - tg) * 1e3
        self.ms += ms
        self.calls += 1
        self.bytes += 4 * (self.B + 1) * self.world + 5 * total   # what arrives at the root: every rank's index row + 5 bytes per entry, its own part included
        return ms

    def verify(self, d_off, d_xs, d_df):
        """Did the root receive what the ranks produced?  To be called right behind run(), before anything rewrites the
        ranks' arrays, outside every timed region: every rank digests its own (offsets, xs, diff), the digests travel
        over torch.distributed, rank 0 digests the segment it holds for every rank.  True / False on rank 0, None elsewhere."""
        _device_sync()
        mine = torch.tensor(payload_digest(d_off, d_xs, d_df), dtype=torch.int64, device=self.cdev)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        if self.world > 1:
            self.dist.all_gather(every, mine)
        else:
            every = [mine]
        if self.rank != 0 or self.root is None:
            return None
        index, xs_all, df_all = self.root
        ok, at = True, 0
        for r in range(self.world):
            pn = int(every[r][0].item())
            got = payload_digest(index[r], xs_all[at:], df_all[at:]) if at + pn <= xs_all.numel() else None
            ok = ok and got == [int(v) for v in every[r].tolist()]
            at += pn
        return bool(ok)


def form_group(core, dist, world, rank, local_rank, cdev, rehearse):
    """This process's core joins a group of `world` ranks below the C-ABI (mi355_group_adopt_rank, csrc/group.hip: RCCL)
    with an id rank 0 makes and torch.distributed hands around.  Returns (group or None, text for config.gather_impl).
    Every rank takes part in every collective of this function whatever happened to it before: a rank that raised and
    skipped a broadcast would leave the others waiting in it (torch.distributed's timeout is half an hour); and every
    rank leaves it the same way (all with a group, or none)."""
    impl = ("mi355_group_gather (csrc/group.hip) over the tests' inter-process stand-in for RCCL: REHEARSAL"
            if rehearse else "mi355_group_gather (RCCL, csrc/group.hip)")
    group = None
    stage, err = "mi355_group_unique_id", None
    ident = np.zeros(1 + 128, np.uint8)          # [0] = 1: rank 0 made the id
    if rank == 0:
        try:
            from cudavideostream_amd.group import unique_id
            ident[1:] = unique_id()
            ident[0] = 1
        except Exception as e:   # noqa: BLE001
            err = e
    t_id = torch.from_numpy(ident).to(cdev)
    dist.broadcast(t_id, src=0)
    ident = t_id.cpu().numpy()
    if err is None and ident[0] != 1:
        stage, err = "mi355_group_unique_id on rank 0", RuntimeError("rank 0 could not make the group's id")
    if err is None:
        stage = "mi355_group_adopt_rank"
        try:
            from cudavideostream_amd.group import CUDAGroup
            group = CUDAGroup.adopt(core, world, rank, ident[1:])
        except Exception as e:   # noqa: BLE001
            group, err = None, e
    if err is not None:
        impl = f"torch.distributed (mi355_group unavailable: {repr(err)[:100]})"
        # WHICH step failed, on WHICH rank, with the library's own text (it names the RCCL entry point:
        # csrc/group.hip RCCL_TRY) -- before anything else happens to this process
        print(f"bench.py: rank {rank} of {world} (device {local_rank}): forming the group failed in {stage}: {err!r}",
              file=sys.stderr, flush=True)
    ok = torch.tensor([1 if group is not None else 0], device=cdev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank takes the same way
    if int(ok.item()) == 0 and group is not None:
        group.close()
        group = None
        impl = "torch.distributed (mi355_group unavailable on another rank)"
    return group, impl


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    # stdout carries exactly ONE line (the JSON record): anything libraries print while the job runs
    # (RCCL prints a version banner on stdout at communicator creation) is routed to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0   # every rank on the one GPU
        if "MI355_RCCL_LIB" not in os.environ or "MOCK_RCCL_SHM" not in os.environ:
            raise SystemExit("bench.py --rehearse-on-one-gpu under a launcher of your own: export MI355_RCCL_LIB="
                             "tests/mock_rccl/librccl_mock_ipc.so and MOCK_RCCL_SHM=/some_name for all ranks "
                             "(`python bench.py --gpus N --rehearse-on-one-gpu` does it by itself)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # tensors torch.distributed moves: on the device with RCCL, on the host with gloo (the rehearsal)
    cdev = torch.device("cpu") if rehearse else dev
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run (also at N = 1)
        import torch.distributed as dist
        import datetime
        # (a rank that dies inside a collective must not hold the others for torch.distributed's default half an hour)
        patience = datetime.timedelta(minutes=8)
        if rehearse:
            dist.init_process_group("gloo", timeout=patience)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=patience)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    W, H, B, K = args.width, args.height, args.batch, args.steps
    n = 3 * W * H
    rr = args.shard == "roundrobin"
    smi_before = smi_snapshot() if rank == 0 else None
    if rr:   # local frame k is global frame rank + k*world; its predecessor travels with it
        mine = gx.roundrobin_frames(rank, world, B * world)
        base = synth.webcam_frame(-1, W, H, seed=21, device=dev)
        frames = torch.stack([synth.webcam_frame(t, W, H, seed=21, device=dev) for t in mine])
        prevs = torch.stack([synth.webcam_frame(t - 1, W, H, seed=21, device=dev) for t in mine])
    else:
        base, frames = synth.webcam_stream(B, W, H, seed=21 + rank, device=dev)
    cap = max(B * n // 8, 1 << 20)
    d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
    d_df = torch.empty(cap, dtype=torch.uint8, device=dev)

    # the core runs on its OWN stream: consecutive batches are then pipelined inside the library (csrc/core.hip,
    # run_batch); the timed region ends with a device-wide synchronisation
    # (a process that also holds torch.distributed / RCCL streams: the core's streams in a priority class of their own, or
    # they share hardware queues with the framework's stream pools and batches no longer overlap: include/mi355diff.h)
    from cudavideostream_amd import lib as _L
    core_flags = _L.FLAG_OWN_QUEUES if (args.own_queues == "1" or (args.own_queues == "auto" and (world > 1 or "RANK" in os.environ))) else 0
    core = CUDACore(W, H, max_batch=B, device=local_rank, flags=core_flags)
    core.set_state(base.cpu().numpy())
    torch.cuda.synchronize()   # the synthetic frames were made on torch's stream

    # The exchange step below the C-ABI (mi355_group_gather: RCCL all-gather of counts + point-to-point sends to
    # rank 0, csrc/group.hip): this process's core joins a group of `world` ranks with an id rank 0 makes and
    # torch.distributed hands around.  If the group cannot be formed (a harness matter, not the path's), the
    # torch.distributed form of the same exchange (cudavideostream_amd/gather.py) is used and named in the line.
    group, gather_impl = None, "n/a"
    # (under a launcher the group is formed at N = 1 too: the line then carries ranks_seen / gather_ms / gather_bytes of
    # the same code path the N > 1 runs take)
    if dist is not None and args.gather != "none":
        group, gather_impl = form_group(core, dist, world, rank, local_rank, cdev, rehearse)
        # Without the group: where the exchange lies INSIDE the timed steps (--gather last / every / index) a scaling line
        # must measure the path's own exchange -- no silent change of what is timed: exit 3 unless --allow-gather-fallback
        # (the torch.distributed form then runs on torch's stream, and so do the batches in front of it).  With the
        # default (--gather after: the exchange BEHIND the timed steps) the steps run exactly as they do with a group -- on
        # the core's own stream, pipelined -- and the torch.distributed form of the exchange runs behind them, timed the
        # same way (final_gather_ms, gather_ran); config.gather_impl names it with the library's error.
        if group is None and (rehearse or (args.gather != "after" and not args.allow_gather_fallback)):
            print(f"bench.py: rank {rank}: the RCCL group could not be formed ({gather_impl}); "
                  f"--allow-gather-fallback measures the torch.distributed form instead", file=sys.stderr, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)
        if group is None and args.gather != "after":
            core.use_torch_stream()   # the torch.distributed form of the exchange runs on torch's stream, between the batches

    xch = Exchange(group, dist, world, rank, B, cap, dev, cdev) if dist is not None and args.gather != "none" else None
    gather_verified = None

    def exchange_payload():
        xch.run(d_off, d_xs, d_df)

    def step(last):
        if rr:
            core.diff_pairs_batch(frames, prevs, B, d_off, d_xs, d_df, cap)
        else:
            core.diff_stream_batch(frames, B, d_off, d_xs, d_df, cap)
        if dist is not None and args.gather != "none":
            # the path itself has no collective (independent streams); the one exchange step is the gather
            # of the changed-pixel stream to rank 0: of the final batch -- behind the timed steps ("after", the default:
            # not here) or as part of the last one ("last") --, of every batch ("every"), or the per-frame index every step
            # and the payload at the end ("index")
            if args.gather == "every" or (args.gather in ("last", "index") and last):
                exchange_payload()
            elif args.gather == "index":
                core.synchronize()   # d_off is written on the core's side stream; torch.distributed runs on torch's
                gx.gather_index(d_off, dst=0)

    def plain_step(_last=False):
        if rr:
            core.diff_pairs_batch(frames, prevs, B, d_off, d_xs, d_df, cap)
        else:
            core.diff_stream_batch(frames, B, d_off, d_xs, d_df, cap)

    def timed_window(k, fn):
        """EXACTLY k steps bracketed by a barrier + device synchronisation on both sides, MAX over the ranks (seconds)."""
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            fn(i == k - 1)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # The chip's state.  A chip that has been idle needs ~12 ms of this load to reach its clocks, then runs the path at
    # 0.95-0.99 of the roofline while it is cool, dips ~0.2 s into the load and settles at its sustained level (0.91-0.92
    # over 18 s: profiles/r05ao_* ... r05as_*).  W + K steps as the first GPU work of a process (W = 5, K = 20: 12 ms) measure
    # that ramp, not the path: they are kept as `cold_start_window` (what rounds 1-5 printed as `value`).  The line's
    # `value` is the same W + K steps behind --preheat-s seconds (default 3) of the same step, untimed: the state a stream
    # that runs for longer than a few seconds is in.  (`steady_state`, 1000 more steps behind the timed ones, stays as well.)
    cold = None
    preheat_steps = 0
    if args.preheat_s > 0:
        for _ in range(args.warmup):
            plain_step()
        cold = timed_window(K, plain_step)
        tp = time.perf_counter()
        while time.perf_counter() - tp < args.preheat_s:
            for _ in range(64):
                plain_step()
            preheat_steps += 64
            core.synchronize()

    for i in range(args.warmup):
        step(i == args.warmup - 1)   # the last warm-up step also runs the exchange (RCCL sets up its peer channels on first use)
    if xch is not None and args.gather == "after":
        exchange_payload()           # ... in every mode
    torch.cuda.synchronize()
    core.set_timing(True)
    core.reset_timing()
    if xch is not None:
        xch.reset()
    elapsed = timed_window(K, step)

    ms_pack, ms_scan, ms_expand, launches = core.get_kernel_timing()
    core.set_timing(False)
    # (the batches ran on the core's own stream, pipelined, unless an exchange in its torch.distributed form lies between them)
    pipelined = core.get_option(1) == 1 and (group is not None or dist is None or args.gather in ("none", "after"))
    if xch is not None and args.gather != "after" and xch.calls:
        gather_verified = xch.verify(d_off, d_xs, d_df)   # the gather inside the last timed step: checked before anything rewrites the arrays
    # The job's ONE exchange ("after"): the final batch's changed-pixel stream of every rank to rank 0, right behind the K
    # timed steps (barrier + device synchronisation either side, MAX over the ranks like the steps' time).  It is the
    # epilogue of a stream, not a step of the hot path: 5 bytes per changed byte of a whole batch over ONE xGMI link per
    # rank (187 MB at 1080p: ~3 ms, against 0.5 ms per step) would be a quarter of a 20-step window and nothing of an
    # hour of video.  `value` is the K steps; `value_with_final_gather` has the exchange in the denominator.
    gather_after_s = None
    if xch is not None and args.gather == "after":
        gather_after_s = timed_window(1, lambda _l: exchange_payload())
        gather_verified = xch.verify(d_off, d_xs, d_df)   # the ranks' digests of THIS batch, before any further step
    g_last = ({"ms": xch.ms, "calls": xch.calls, "bytes": xch.bytes, "ranks_seen": xch.ranks_seen} if xch is not None
              else {"ms": 0.0, "calls": 0, "bytes": 0, "ranks_seen": world if world > 1 else 1})
    # What the path SUSTAINS: the same step `--steady-steps` times (default 1000: half a second), right behind the
    # timed region.
    steady = None
    if args.steady_steps > 0:    # (every rank: the same steps between the same barriers, MAX over the ranks, no exchange)
        steady = (args.steady_steps, timed_window(args.steady_steps, plain_step))
    # secondary measurement (N > 1): the same job with the gather after EVERY batch -- the exchange at its worst
    # (every byte of every rank funnelled to one GPU), so that the scaling curve shows what the gather costs
    g_every = None
    if world > 1 and group is not None and args.gather_every_steps > 0:
        xch.reset()
        K2 = args.gather_every_steps

        def step_and_gather(_last):
            plain_step()
            exchange_payload()
        e2 = timed_window(K2, step_and_gather)
        every_ok = xch.verify(d_off, d_xs, d_df)
        g_every = {"value": round(world * B * K2 / e2, 1), "unit": "frames/s", "steps": K2,
                   "ms_per_step": round(e2 / K2 * 1e3, 4), "gather_ms": round(xch.ms / max(xch.calls, 1), 4),
                   "gather_bytes": xch.bytes // max(xch.calls, 1),
                   "gather_gbps": round(xch.bytes / max(xch.ms, 1e-9) / 1e6, 1), "gather_verified": every_ok,
                   "note": "every batch's changed-pixel stream of every rank gathered to rank 0 (rank 0's own timing of "
                           "mi355_group_gather, host synchronised either side)"}
        if rank == 0 and every_ok is False:
            gather_verified = False
    core.synchronize()
    off = d_off.cpu().numpy().view(np.uint32)
    p_total = int(off[-1])
    assert p_total <= cap, "output capacity too small for this stream"

    # BASELINE configs[4] across the ranks of this job (whenever a launcher started it: N > 1, or N = 1 with real RCCL)
    c5 = None
    if dist is not None and not args.no_config5 and not rr:
        c5 = config5_across_ranks(args, dist, world, rank, local_rank, dev, cdev, rehearse)

    if rank == 0:
        alg_bytes = 2.0 * n * B + 5.0 * p_total          # B_alg = 2N + 5P per frame
        pmc = None if rr else pmc_counters(B, W, H)
        out = {
            "metric": metric_name(W, H),
            "value": round(world * B * K / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            **({"rehearsal": "NOT A MEASUREMENT: every rank ran on device 0 (torch.distributed over gloo, the exchange "
                             "through tests/mock_rccl/librccl_mock_ipc.so, payloads staged through host shared memory)"}
               if rehearse else {}),
            "config": {"workload": (f"{W}x{H} BGR24 S1 webcam sequence dealt round-robin to the ranks, {B}-frame "
                                    f"batches resident in HBM, stateless diff against the raw predecessor"
                                    f"+threshold(20)+pack, ordered output" if rr else
                                    f"{W}x{H} BGR24 S1 webcam stream, {B}-frame batches resident in HBM, "
                                    f"stateful diff+threshold(20)+pack, ordered output"),
                       "frames_per_step": B, "changed_bytes_per_frame": round(p_total / B, 1),
                       "parallelism": (f"frames round-robin over {world} ranks" if rr else
                                       f"{world} independent streams" if world > 1 else "1 stream"),
                       "core_flags": ("MI355_FLAG_OWN_QUEUES: the core's streams in the least stream-priority class (hardware queues of "
                                      "their own beside torch.distributed's / RCCL's stream pools); GPU_MAX_HW_QUEUES=%s"
                                      % os.environ.get("GPU_MAX_HW_QUEUES") if core_flags else "0"),
                       "gather": ({"after": "after: one gather-v of the final batch to rank 0 behind the K timed steps (final_gather_ms, "
                                            "value_with_final_gather); --gather last puts it inside the timed region",
                                   "last": "last: one gather-v of the final batch to rank 0 inside the timed region"}.get(args.gather, args.gather)
                                  if dist is not None else "n/a"), "gather_impl": gather_impl},
            # (the K steps + the job's one exchange, MAX over the ranks: `value` has only the steps in its denominator)
            **({"value_with_final_gather": round(world * B * K / (elapsed + gather_after_s), 1),
                "final_gather_ms": round(gather_after_s * 1e3, 4)} if gather_after_s is not None else {}),
            "ranks_seen": g_last["ranks_seen"],
            "gather_ms": round(g_last["ms"] / g_last["calls"], 4) if g_last["calls"] else None,
            "gather_bytes": g_last["bytes"] // g_last["calls"] if g_last["calls"] else None,
            "gather_ran": bool(g_last["calls"]) if dist is not None else None,
            "gather_every": g_every,
            "gather_verified": gather_verified,
            **({"config5": c5} if c5 is not None else {}),
            "roofline": path_roofline(alg_bytes, (ms_pack, ms_scan, ms_expand), launches, rr, pmc,
                                      wall_ms=elapsed / K * 1e3 if pipelined else None),
        }
        if steady is not None:
            k3, s3 = steady
            out["steady_state"] = {"steps": k3, "seconds": round(s3, 3), "frames_per_s": round(world * B * k3 / s3, 1),
                                   "ms_per_step": round(s3 / k3 * 1e3, 4),
                                   "achieved_gbps": round(alg_bytes / (s3 / k3) / 1e9, 1),   # per GPU (rank 0's bytes)
                                   "frac": round(alg_bytes / (s3 / k3) / 1e9 / HBM_PEAK_GBPS, 4),
                                   **fractions(alg_bytes, pmc["hbm_bytes_per_launch"] if pmc else None, s3 / k3, rr),
                                   "note": "the same step repeated right behind the K timed steps, wall clock between device "
                                           "synchronisations: what the path sustains (DESIGN.md section 8)"}
        if cold is not None:
            out["cold_start_window"] = {"value": round(world * B * K / cold, 1), "ms_per_step": round(cold / K * 1e3, 4),
                                        "frac": round(alg_bytes / (cold / K) / 1e9 / HBM_PEAK_GBPS, 4),
                                        "note": f"the same {args.warmup} warm-up + {K} timed steps as the FIRST GPU work of the process "
                                                "(what rounds 1-5 printed as `value`): it sits on the chip's clock ramp"}
        out["preheat"] = {"seconds": args.preheat_s, "steps": preheat_steps,
                          "note": "the same step, untimed, in front of the warm-up + timed steps whose rate is `value`: a chip that "
                                  "has been idle needs ~12 ms of load to reach its clocks and a second or two to settle "
                                  "(profiles/r05ao-r05as); --preheat-s 0: none"}
        if world == 1 and not args.no_pair and not rr:
            out["pair_mode"] = pair_mode(args, core, frames, d_off, d_xs, d_df, cap, n)
            out["regimes"] = regimes(args, dev)
        if world == 1 and not args.no_pair and not rr:
            out["two_streams_one_gpu"] = two_streams(args, core, frames, base, d_off, d_xs, d_df, cap, dev)
        if world == 1 and not args.no_filters and not rr:
            out.update(filter_configs(args, dev))
        if world == 1 and not args.no_config5 and not rr and (W, H) == (1920, 1080):
            out["config5_per_gpu"] = config5_per_gpu(args, dev)
        if world == 1 and not args.no_host_path:
            out["host_path"] = host_path(args, base, frames)
        out["board"] = board_fingerprint(core, "right after the timed region")
        out["board"]["rocm_smi_before"] = smi_before
        out["board"]["rocm_smi_after"] = smi_snapshot()
        stream_gbps = out["board"].get("hbm_stream_read_gbps")
        if stream_gbps:
            for blk in (out["roofline"], out.get("steady_state")):
                if blk and blk.get("actual_gbps"):
                    blk["frac_of_achievable"] = round(blk["actual_gbps"] / stream_gbps, 4)
        parity_failed = False
        if world == 1 and not args.no_cpu and not rr:
            out["cpu_baseline"], headline_ok = cpu_baseline(args, base, frames, dev)
            # parity next to every number of the line (product vs oracle, bit-exact, outside the timed regions)
            par = {"headline": headline_ok}
            if not args.no_pair or not args.no_filters:
                par.update(parity_of_secondary_lines(args, dev, frames))
            if isinstance(out.get("config5_per_gpu"), dict) and "parity" in out["config5_per_gpu"]:
                par["config5_per_gpu"] = out["config5_per_gpu"]["parity"]
            out["parity"] = par
            parity_failed = not all(par.values())
        else:
            out["cpu_baseline"] = None
        if c5 is not None and "parity" in c5:
            out["parity"] = dict(out.get("parity") or {}, config5=c5["parity"])
            parity_failed = parity_failed or c5["parity"] is not True
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
        if parity_failed:
            print(f"bench.py: PARITY FAILED: {out.get('parity')}", file=sys.stderr)
            core.close()
            raise SystemExit(4)
        if gather_verified is False or (c5 is not None and c5.get("gather_verified") is False):
            print("bench.py: GATHER NOT VERIFIED: the root's copy of a rank's batch differs from what that rank produced",
                  file=sys.stderr)
            core.close()
            raise SystemExit(5)
    if group is not None:
        group.close()
    core.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def host_path(args, base, frames, reps=60):
    """Secondary line, never `value`: the reference's per-frame entry point exec_core (kernels.cu:430-525)
    through mi355_exec with pinned host buffers -- H2D of the frame, kernels, D2H of count/diff/xs and the
    two synchronisations of the reference.  PCIe/latency-bound."""
    n = base.numel()
    with CUDACore(args.width, args.height, sample_mat_data=base.cpu().numpy()) as c:
        h_frame, n_frame, o_frame, h_xs = CUDACore.alloc_arrays(args.height, args.width)
        src = frames[:min(frames.shape[0], 16)].cpu().numpy()
        for i in range(5):
            h_frame.array[:n] = src[i % src.shape[0]]
            c.exec_core(h_frame.array, None, "", h_xs.array)
        copy_s = 0.0
        t0 = time.perf_counter()
        for i in range(reps):
            tc = time.perf_counter()
            h_frame.array[:n] = src[i % src.shape[0]]   # stands in for the capture thread's write
            copy_s += time.perf_counter() - tc
            c.exec_core(h_frame.array, None, "", h_xs.array)
        dt = time.perf_counter() - t0 - copy_s
        for a in (h_frame, n_frame, o_frame, h_xs):
            a.free()
        # the same frames with several in flight (mi355_pipe_*): upload of frame k+1 beside the kernels of
        # frame k, no host synchronisation between the pack and the way back.  One pinned buffer set per
        # resident frame, depth of them in flight.
        depth, preps = 4, 4 * reps
        from cudavideostream_amd.core import PinnedArray
        nsrc = src.shape[0]
        ring = [(PinnedArray(n + 32), PinnedArray(4 * n + 32, np.int32)) for _ in range(nsrc)]
        for k in range(nsrc):
            ring[k][0].array[:n] = src[k]
        c.set_state(base.cpu().numpy())
        c.pipe_open(depth)
        tickets = [None] * nsrc
        fill_s = 0.0

        def finish(k):
            # the diff bytes overwrote the head of the frame (in/out buffer, kernels.cu:461,522): put the
            # frame back, which in the server is the capture thread's job
            nonlocal fill_s
            pos = c.exec_wait(tickets[k])
            tickets[k] = None
            tc = time.perf_counter()
            ring[k][0].array[:pos] = src[k][:pos]
            fill_s += time.perf_counter() - tc

        t0 = time.perf_counter()
        for i in range(preps):
            if i >= depth:
                finish((i - depth) % nsrc)
            k = i % nsrc
            tickets[k] = c.exec_submit(ring[k][0].array, None, "", ring[k][1].array)
        for i in range(preps, preps + depth):
            finish((i - depth) % nsrc)
        dt_pipe = time.perf_counter() - t0 - fill_s
        c.pipe_close()
        for bufs in ring:
            for a in bufs:
                a.free()
    return {"frames_per_s": round(reps / dt, 1), "ms_per_frame": round(dt / reps * 1e3, 4),
            "note": "mi355_exec per frame, blocking: H2D frame + kernels + count/diff/xs stored to the pinned buffers, 1 sync (PCIe-inclusive)",
            "pipelined_frames_per_s": round(preps / dt_pipe, 1), "pipeline_depth": depth,
            "pcie_h2d_gbps": round(preps * n / dt_pipe / 1e9, 2)}


RAMP_MS = 15.0   # a chip that has been idle needs ~12 ms of load to reach its clocks (profiles/r05ar_*)


def warm_up(core, fn, warm):
    """The SECONDARY lines' warm-up: `warm` calls, then as many more as it takes to have the chip under this load for
    RAMP_MS (their timed regions are a few milliseconds long: without this they would measure the clock ramp.  The
    headline's warm-up is the driver's --warmup, not this)."""
    def sync():
        core.synchronize()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(warm, 1)):
        fn()
    sync()
    for _ in range(2):
        el = time.perf_counter() - t0
        if el * 1e3 >= RAMP_MS:
            break
        per = el / max(warm, 1)
        for _ in range(min(int((RAMP_MS * 1e-3 - el) / max(per, 1e-6)) + 1, 5000)):
            fn()
        sync()


def timed_path(core, fn, reps, warm=3):
    """Runs fn() reps times with the core's kernel timers on: (seconds per call, (pack, scan, expand) ms sums, launches)."""
    core.set_timing(True)
    warm_up(core, fn, warm)
    core.reset_timing()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    a, b, c, launches = core.get_kernel_timing()
    core.set_timing(False)
    return dt / reps, (a, b, c), launches


def wall_per_call(core, fn, reps, warm=3):
    """Seconds per fn() on the core's OWN stream (consecutive batches pipelined inside the library, as the headline
    runs), wall clock between device synchronisations."""
    warm_up(core, fn, warm)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    core.synchronize()
    return (time.perf_counter() - t0) / reps


def path_line(B, n, p, sec_per_call, ms, launches, pair, sec_pipelined=None):
    """Secondary line of one regime: whole-path rate and roofline fraction, ALL kernels in the denominator.
    `frac_sequential` / `all_kernels_ms` / `kernels_us`: one batch after the other on a caller's stream, the kernels' own
    durations (HIP events) added up.  `frac` / `ms_per_launch`: the same calls on the core's own stream, wall clock per
    call -- the way the headline is measured (batches pipelined inside the library)."""
    alg = 2.0 * n * B + 5.0 * p
    r = path_roofline(alg, ms, launches, pair)
    out = {"frames_per_s": round(B / sec_per_call, 1), "frames_per_launch": B,
           "changed_bytes_per_frame": round(p / B, 1), "all_kernels_ms": r["kernel_ms"],
           "achieved_gbps": r["achieved"], "frac": r["frac"],
           "kernels_us": [k["avg_us"] for k in r["kernels"]]}
    if sec_pipelined is not None:
        # ONE method for `frac`, the headline's: the calls on the core's OWN stream, wall clock per call (the library's
        # default way of running: batches overlapped unless its adaptive overlap sees dense input and runs them one after
        # the other by itself).  The caller's-stream figure (kernels' own durations added up) stays under its own keys.
        gbps = alg / sec_pipelined / 1e9
        out.update({"frac_sequential": r["frac"], "achieved_gbps_sequential": r["achieved"],
                    "frames_per_s_sequential": out["frames_per_s"],
                    "frac": round(gbps / HBM_PEAK_GBPS, 4), "achieved_gbps": round(gbps, 1),
                    "frames_per_s": round(B / sec_pipelined, 1), "ms_per_launch": round(sec_pipelined * 1e3, 4),
                    "basis": "frac / frames_per_s / ms_per_launch: the core's own stream, wall clock per call (as the "
                             "headline); *_sequential, all_kernels_ms, kernels_us: a caller's stream, HIP-event "
                             "durations of the three kernels added up"})
    return out


def pair_mode(args, core, frames, d_off, d_xs, d_df, cap, n):
    """Secondary line: stateless frame pairs with NO reuse between the two operands (cur = first half
    of the resident frames, prev = second half), i.e. 2N bytes of HBM reads per frame -- the plain
    streaming rate of the same kernels.  (The halves show different rectangle positions, so P is larger
    than in the stream.)"""
    B = frames.shape[0] // 2
    cur, prev = frames[:B], frames[B:2 * B]
    core.use_torch_stream()   # one batch after the other (a caller's stream is never pipelined): the kernels' own times add up
    sec, ms, launches = timed_path(core, lambda: core.diff_pairs_batch(cur, prev, B, d_off, d_xs, d_df, cap), 20)
    core.use_own_stream()
    sec_pipe = wall_per_call(core, lambda: core.diff_pairs_batch(cur, prev, B, d_off, d_xs, d_df, cap), 20)
    p = int(d_off.cpu().numpy().view(np.uint32)[B])
    return path_line(B, n, p, sec, ms, launches, True, sec_pipe)


def two_streams(args, core, frames, base, d_off, d_xs, d_df, cap, dev, reps=20):
    """Secondary line, NOT the bench's workload: two independent streams of the same shape on the one GPU, each on its
    own core and stream.  Their kernels overlap in every combination, which a single stream's batches cannot (the
    pack kernel of batch k + 1 needs the state of batch k): the aggregate shows how far instruction issue, not the
    memory system, is from being saturated by one stream."""
    B = frames.shape[0]
    try:
        base2, frames2 = synth.webcam_stream(B, args.width, args.height, seed=121, device=dev)
        o2 = (torch.zeros_like(d_off), torch.empty_like(d_xs), torch.empty_like(d_df))
        with CUDACore(args.width, args.height, max_batch=B) as core2:
            core2.set_state(base2.cpu().numpy())
            torch.cuda.synchronize()

            def both():
                core.diff_stream_batch(frames, B, d_off, d_xs, d_df, cap)
                core2.diff_stream_batch(frames2, B, *o2, cap)

            warm_up(core, both, 3)
            core2.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                both()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            p = int(d_off.cpu().numpy().view(np.uint32)[B]) + int(o2[0].cpu().numpy().view(np.uint32)[B])
        alg = 2.0 * frames.shape[1] * 2 * B + 5.0 * p
        return {"frames_per_s": round(2 * B / dt, 1), "ms_per_round_of_two_batches": round(dt * 1e3, 4),
                "frac": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 4),
                "note": "two cores, two streams, 2 x %d frames per round; wall clock" % B}
    except Exception as e:   # noqa: BLE001  -- a secondary line must not cost the bench line (memory on a shared box)
        return {"skipped": repr(e)[:160]}


def config5_per_gpu(args, dev, ranks=8, B=64, W=3840, H=2160, reps=10):
    """BASELINE configs[4] as ONE GPU of the 8 sees it: a 4K S1 sequence dealt round-robin over 8 ranks, this rank's
    64-frame shard (frames 0, 8, 16, ...) diffed against their raw predecessors, stateless (mi355_diff_pairs_batch;
    the operands share no frame, so both are read with non-temporal loads).  Same measurement as `pair_mode`; parity of
    the first and the last pair of the shard against the oracle."""
    from oracle import pyoracle as po
    n = 3 * W * H
    mine = gx.roundrobin_frames(0, ranks, B * ranks)
    try:
        frames = torch.stack([synth.webcam_frame(t, W, H, seed=31, device=dev) for t in mine])
        prevs = torch.stack([synth.webcam_frame(t - 1, W, H, seed=31, device=dev) for t in mine])
        cap = max(B * n // 8, 1 << 20)
        d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
        d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
        with CUDACore(W, H, max_batch=B) as core:
            torch.cuda.synchronize()
            core.use_torch_stream()
            sec, ms, launches = timed_path(core, lambda: core.diff_pairs_batch(frames, prevs, B, d_off, d_xs, d_df, cap), reps)
            core.use_own_stream()
            sec_pipe = wall_per_call(core, lambda: core.diff_pairs_batch(frames, prevs, B, d_off, d_xs, d_df, cap), reps)
            off = d_off.cpu().numpy().view(np.uint32)
            p = int(off[B])
            assert p <= cap, "config5_per_gpu: output capacity too small"
            out = path_line(B, n, p, sec, ms, launches, True, sec_pipe)
            ok = True
            for t in (0, B - 1):
                cnt, xs, df, _ = po.diff_pack(frames[t].cpu().numpy(), prevs[t].cpu().numpy())
                ok = ok and int(off[t + 1] - off[t]) == cnt and np.array_equal(d_xs[int(off[t]):int(off[t + 1])].cpu().numpy(), xs) \
                    and np.array_equal(d_df[int(off[t]):int(off[t + 1])].cpu().numpy(), df)
        out.update({"workload": f"BASELINE configs[4], one GPU's share: {W}x{H} BGR24 S1 sequence dealt round-robin over {ranks} "
                                f"ranks, rank 0's {B}-frame shard, stateless diff against the raw predecessor + threshold(20) + pack",
                    "parity": bool(ok)})
        return out
    except Exception as e:   # noqa: BLE001  -- a secondary line must not cost the bench line (memory on a shared box)
        return {"skipped": repr(e)[:160]}


def config5_across_ranks(args, dist, world, rank, local_rank, dev, cdev, rehearse):
    """BASELINE configs[4] as the job itself: ONE 3840x2160 S1 sequence dealt round-robin over the ranks that are actually
    there (rank r takes frames r, r + world, ...: a 64-frame shard each), every frame diffed against its raw predecessor,
    stateless (mi355_diff_pairs_batch: the shards need nothing of each other), then the path's one exchange: gather-v of
    every rank's changed-pixel stream to rank 0 over RCCL (mi355_group_gather on a group of its own).  Timed like the
    headline: K passes over the shard between barriers, MAX over the ranks; the gather behind them, timed on its own.
    Parity: rank 0 looks at ITS copy of the gathered streams -- for every rank the first and the last frame of that rank's
    shard -- against the oracle's diff of the regenerated frames: kernel, gather and the round-robin bookkeeping in one check.
    Every rank takes part in every collective here; only rank 0's return value is used."""
    W, H, B = args.config5_size
    if rehearse and B > 16:   # the stand-in stages payloads through 256 MB of shared host memory: a quarter of the shard is enough to rehearse
        B = 16
    K5 = max(args.config5_steps, 1)
    n = 3 * W * H
    seed = 31
    res = {"workload": f"BASELINE configs[4]: {W}x{H} BGR24 S1 sequence of {B * world} frames dealt round-robin over {world} rank(s), "
                       f"{B}-frame shards resident in HBM, stateless diff against the raw predecessor + threshold(20) + pack, "
                       f"then ONE gather-v of all shards' streams to rank 0",
           "frames_per_rank": B, "steps": K5}
    # every rank makes its shard; a rank that cannot (memory on a shared box) must not leave the others in the collectives below:
    # the ranks agree first
    mine = gx.roundrobin_frames(rank, world, B * world)
    core, why = None, ""
    try:
        frames = torch.stack([synth.webcam_frame(t, W, H, seed=seed, device=dev) for t in mine])
        prevs = torch.stack([synth.webcam_frame(t - 1, W, H, seed=seed, device=dev) for t in mine])
        cap = max(B * n // 8, 1 << 20)
        d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
        d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
        from cudavideostream_amd import lib as _L
        core = CUDACore(W, H, max_batch=B, device=local_rank, flags=_L.FLAG_OWN_QUEUES)
        torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001
        why = repr(e)[:160]
        print(f"bench.py: rank {rank}: config5 could not be set up: {why}", file=sys.stderr, flush=True)
    ready = torch.tensor([1 if core is not None else 0], device=cdev)
    dist.all_reduce(ready, op=dist.ReduceOp.MIN)
    if int(ready.item()) == 0:
        if core is not None:
            core.close()
        return {"skipped": f"a rank could not set its shard up ({why or 'another rank'})"} if rank == 0 else None
    group, impl = form_group(core, dist, world, rank, local_rank, cdev, rehearse)
    res["gather_impl"] = impl
    xch = Exchange(group, dist, world, rank, B, cap, dev, cdev)

    def step():
        core.diff_pairs_batch(frames, prevs, B, d_off, d_xs, d_df, cap)

    warm_up(core, step, 3)
    xch.run(d_off, d_xs, d_df)     # RCCL sets up this communicator's peer channels on first use (host-synchronised: the chip idles)
    xch.reset()
    warm_up(core, step, 3)         # ... so the chip is put under this load again right in front of the timed passes
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K5):
        step()
    core.synchronize()
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    p = int(d_off[B].item()) & 0xFFFFFFFF
    assert p <= cap, "config5: output capacity too small"
    # per rank: its own seconds for the K passes, its changed bytes
    mine_t = torch.tensor([mine_s, float(p)], dtype=torch.float64, device=cdev)
    every = [torch.zeros_like(mine_t) for _ in range(world)]
    if world > 1:
        dist.all_gather(every, mine_t)
    else:
        every = [mine_t]
    # the exchange, behind the timed passes (barrier + synchronisation either side, MAX over the ranks)
    dist.barrier()
    torch.cuda.synchronize()
    tg = time.perf_counter()
    xch.run(d_off, d_xs, d_df)
    dist.barrier()
    eg = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=cdev)
    dist.all_reduce(eg, op=dist.ReduceOp.MAX)
    gather_s = float(eg.item())
    verified = xch.verify(d_off, d_xs, d_df)
    if rank == 0:
        from oracle import pyoracle as po
        index, xs_all, df_all = xch.root
        seg = gx.roundrobin_order(index, B * world)     # global frame t -> its entries inside the rank-major gathered arrays
        ok = True
        for r in range(world):
            for k in sorted({0, B - 1}):
                t = r + k * world
                cur = synth.webcam_frame(t, W, H, seed=seed, device=dev).cpu().numpy()
                prv = synth.webcam_frame(t - 1, W, H, seed=seed, device=dev).cpu().numpy()
                cnt, xs, df, _ = po.diff_pack(cur, prv)
                a, b = seg[t]
                ok = ok and b - a == cnt and np.array_equal(xs_all[a:b].cpu().numpy(), xs) and np.array_equal(df_all[a:b].cpu().numpy(), df)
        fr = [2.0 * n * B + 5.0 * float(e[1].item()) for e in every]
        res.update({
            "value": round(world * B * K5 / elapsed, 1), "unit": "frames/s", "ms_per_step": round(elapsed / K5 * 1e3, 4),
            "changed_bytes_per_frame": round(sum(float(e[1].item()) for e in every) / (B * world), 1),
            "frac_per_rank": [round(fr[r] / (float(every[r][0].item()) / K5) / 1e9 / HBM_PEAK_GBPS, 4) for r in range(world)],
            "frac": round(sum(fr) / world / (elapsed / K5) / 1e9 / HBM_PEAK_GBPS, 4),
            "final_gather_ms": round(gather_s * 1e3, 4), "gather_ms": round(xch.ms / max(xch.calls, 1), 4),
            "gather_bytes": xch.bytes // max(xch.calls, 1),
            "gather_gbps": round(xch.bytes / max(xch.ms, 1e-9) / 1e6, 1),
            "value_with_final_gather": round(world * B * K5 / (elapsed + gather_s), 1),
            "ranks_seen": xch.ranks_seen, "gather_verified": verified, "parity": bool(ok),
            "basis": "value / ms_per_step / frac: K passes over every rank's shard between barriers, MAX over the ranks (frac: the "
                     "ranks' mean algorithmic 2N + 5P bytes per pass over that time, per GPU, of 8 TB/s; frac_per_rank: each rank's "
                     "own clock); final_gather_ms: the one gather behind them, barrier to barrier, MAX over the ranks; gather_ms / "
                     "gather_gbps: rank 0's own timing of the call and the bytes that arrived at it; parity: rank 0's copy of the "
                     "first and last frame of every rank's shard against the oracle"})
    if group is not None:
        group.close()
    core.close()
    return res if rank == 0 else None


def regimes(args, dev, B=32):
    """Secondary lines: the same path on the other input regimes of SURVEY.md 8d, 32-frame batches, whole-path
    fractions: S0 refrand pairs (the generator of tests/algorithms_benchmarks.cu:4-10, P ~ 0.85 N), every byte
    changed (P = N), nothing changed (P = 0).  The two output arrays come from mi355_alloc_outputs: in the dense regimes
    the expansion is bound by its stores, and whether the index and the value stream overlap in the memory system is a
    property of the pair's placement (205 or 265 us per 32 S0 pairs: include/mi355diff.h); `plain_allocation` repeats the
    S0 line on two torch.empty arrays -- whatever lot this process draws."""
    W, H = args.width, args.height
    n = 3 * W * H
    cap = B * n
    d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    rnd = torch.stack([synth.refrand_frame(n, 100 + t, device=dev) for t in range(2 * B)])
    flip = rnd[:B] ^ 0x80
    out = {}
    with CUDACore(W, H, max_batch=B) as core:
        t0 = time.perf_counter()
        d_xs, d_df, draws = core.alloc_outputs(cap)
        alloc_ms = (time.perf_counter() - t0) * 1e3
        core.use_torch_stream()

        def line(cur, prev, xs, df):
            sec, ms, launches = timed_path(core, lambda: core.diff_pairs_batch(cur, prev, B, d_off, xs, df, cap), 10, 2)
            core.use_own_stream()
            sec_pipe = wall_per_call(core, lambda: core.diff_pairs_batch(cur, prev, B, d_off, xs, df, cap), 10, 2)
            core.use_torch_stream()
            p = int(d_off.cpu().numpy().view(np.uint32)[B])
            return path_line(B, n, p, sec, ms, launches, True, sec_pipe)

        for name, cur, prev in (("S0_refrand_pairs", rnd[B:], rnd[:B]), ("P_eq_N_pairs", flip, rnd[:B]),
                                ("P_eq_0_pairs", rnd[:B], rnd[:B].clone())):
            out[name] = line(cur, prev, d_xs, d_df)
        out["outputs"] = {"from": "mi355_alloc_outputs", "value_arrays_drawn": draws, "ms": round(alloc_ms, 1)}
        t_xs = torch.empty(cap, dtype=torch.int32, device=dev)
        t_df = torch.empty(cap, dtype=torch.uint8, device=dev)
        plain = line(rnd[B:], rnd[:B], t_xs, t_df)
        out["S0_refrand_pairs"]["plain_allocation"] = {k: plain[k] for k in ("frac", "frac_sequential", "kernels_us")}
        core.synchronize()
        core.dev_free(d_xs)
        core.dev_free(d_df)
    return out


def filter_configs(args, dev, B=192, reps=10):
    """BASELINE configs 3 and 4 end to end on a resident batch (what the server does with a frame when the
    visualiser / noise filter is on, kernels.cu:457-520): the visualiser's frame AND the packed diff stream.
      config3: weighted grayscale + histogram + two-max + binarize (one gray byte per pixel kept between the two
               passes), then diff+threshold+pack.  Algorithmic bytes per frame: 3N + 5P, the fused model BASELINE
               names (colour frame read ONCE for visualiser and diff: N, binarized frame out: N, state: N); the
               implementation reads the colour frame twice (the threshold is global: DESIGN.md section 4).
      config4: 3x3 noise filter (N + N), diff+threshold+pack of the filtered frames (2N + 5P), red motion map
               from the packed indices (N cleared + P/3 painted ~ N).
    frac = algorithmic bytes / all kernels of the chain (HIP events on the stream) / 8 TB/s."""
    from cudavideostream_amd import lib as L
    W, H = args.width, args.height
    n = 3 * W * H
    _, frames = synth.webcam_stream(B + 1, W, H, seed=33, device=dev)
    cur = frames[1:]
    vis = torch.empty((B, n), dtype=torch.uint8, device=dev)
    filt = torch.empty((B, n), dtype=torch.uint8, device=dev)
    cap = B * n // 4
    d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
    d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
    g = np.exp(-(np.arange(-1, 2)[:, None] ** 2 + np.arange(-1, 2)[None, :] ** 2) / (2.0 * 1.5 * 1.5))
    k = (g / g.sum()).astype(np.float32).reshape(-1)
    res = {}
    with CUDACore(W, H, k=k, max_batch=B) as core:
        core.set_state(frames[0].cpu().numpy())
        torch.cuda.synchronize()

        def config3():
            core.filter_batch(L.OP_GRAY_WEIGHTED_BINARIZE, cur, vis, B)
            core.diff_stream_batch(cur, B, d_off, d_xs, d_df, cap)

        def config4():
            core.filter_batch(L.OP_CONV3X3, cur, filt, B)
            core.diff_stream_batch(filt, B, d_off, d_xs, d_df, cap)
            core.red_stream_batch(d_off, d_xs, B, vis)

        def wall_us(fn):
            warm_up(core, fn, 3)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            core.synchronize()
            return (time.perf_counter() - t0) * 1e6 / (reps * B)

        # config3's algorithmic bytes: 3N + 5P -- BASELINE names the chain "fused, sharing the diff's loads": colour
        # frame read once (N), binarized frame written (N), state (N), 5P out; config4: 5N + 5P.
        # Timed twice: on the core's OWN stream, the way the library is meant to run (the frame filter of batch k + 1
        # beside the expansion of batch k, include/mi355diff.h), wall clock between device synchronisations; and one
        # batch after the other on a caller's stream (`sequential_us_per_frame`).
        for name, fn, fixed in (("config3", config3, 3.0 * n), ("config4", config4, 5.0 * n)):
            core.use_own_stream()
            us = wall_us(fn)
            core.use_torch_stream()
            us_seq = wall_us(fn)
            p = (int(d_off[B].item()) & 0xFFFFFFFF) / B
            alg = fixed + 5.0 * p
            gbps = alg / (us * 1e-6) / 1e9
            res[name] = {"workload": BASELINE_CONFIGS.get(name, name), "us_per_frame": round(us, 3),
                         "sequential_us_per_frame": round(us_seq, 3),
                         "frames_per_s": round(1e6 / us, 1), "frames_per_launch": B,
                         "changed_bytes_per_frame": round(p, 1), "algorithmic_bytes_per_frame": int(alg),
                         "achieved_gbps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4),
                         "basis": "wall clock per frame on the core's own stream (filters of a batch beside the expansion of "
                                  "the batch before); sequential_us_per_frame: the same calls on a caller's stream"}
        # the 5x5 median the reference evaluated and left out of its server (tests/noise_filter_benchmark/v3.cu): the one
        # kernel here that is bound by arithmetic (packed 16-bit min / max), not by HBM; N read + N written
        core.use_torch_stream()
        us = wall_us(lambda: core.filter_batch(L.OP_MEDIAN5X5, cur, vis, B))
        res["median5x5"] = {"workload": "5x5 median per colour channel, zeros outside the image; not on the server's path",
                            "us_per_frame": round(us, 3), "frames_per_launch": B, "algorithmic_bytes_per_frame": 2 * n,
                            "achieved_gbps": round(2 * n / (us * 1e-6) / 1e9, 1), "frac": round(2 * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                            "bound": "valu: ~43 packed min / max class instructions per output byte (DESIGN.md K6), HBM idle"}
    return res


BASELINE_CONFIGS = {
    "config3": "BASELINE configs[2]: 1080p diff + grayscale-weighted + binarize filter chain, 1xMI355X",
    "config4": "BASELINE configs[3]: 1080p diff + motion heat-map (red) + noise filter, 1xMI355X",
}


if __name__ == "__main__":
    main()
