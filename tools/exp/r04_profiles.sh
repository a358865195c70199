#!/bin/bash
# Round 4's evidence run on the round's final library: everything lands under gpurun_out/ and is condensed into
# profiles/r04_* in the build container afterwards.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r04prof
{
echo "=== bench.py default"; python bench.py --steps 20 --warmup 5 > gpurun_out/r04prof/bench.json 2> gpurun_out/r04prof/bench.err; echo rc=$?
echo "=== bench.py sequential (MI355_PIPELINE=0)"; MI355_PIPELINE=0 python bench.py --steps 20 --warmup 5 --no-cpu --no-host-path > gpurun_out/r04prof/bench_sequential.json 2>/dev/null; echo rc=$?
echo "=== bench.py 4K stream"; python bench.py --steps 20 --warmup 5 --width 3840 --height 2160 --batch 64 --no-cpu --no-host-path --no-filters --no-pair > gpurun_out/r04prof/bench_4k.json 2>/dev/null; echo rc=$?
echo "=== bench.py 4K round-robin pairs (config 5's per-GPU work, one GPU)"; python bench.py --steps 20 --warmup 5 --width 3840 --height 2160 --batch 64 --shard roundrobin --no-cpu --no-host-path --no-filters --no-pair > gpurun_out/r04prof/bench_4k_roundrobin.json 2>/dev/null; echo rc=$?
echo "=== run_profile.sh r04 (kernel trace on bench.py, FETCH/WRITE on diffbench)"; bash profiles/run_profile.sh r04 5 2>&1 | tail -40
echo "=== filters"; bash profiles/run_profile_filters.sh r04 2>&1 | tail -30
echo "=== pair mode 1080p (128 pairs, consecutive frames)"; bash profiles/pmc_fw.sh pair1080 --pairs --batch 128
echo "=== pair mode 4K (64 pairs)"; bash profiles/pmc_fw.sh pair4k --pairs --width 3840 --height 2160 --batch 64
echo "=== pair mode 1080p, operands that share no frame (128 pairs (0,1), (2,3), ...)"; bash profiles/pmc_fw.sh pair1080apart --apart --batch 128
echo "=== pair mode 4K, operands that share no frame (64 pairs)"; bash profiles/pmc_fw.sh pair4kapart --apart --width 3840 --height 2160 --batch 64
echo "=== S0 refrand pairs, 32 frames"; bash profiles/pmc_fw.sh s0 --regime s0 --batch 32
echo "=== P = N pairs, 32 frames"; bash profiles/pmc_fw.sh flip --regime flip --batch 32
echo "=== 4K stream, 64 frames"; bash profiles/pmc_fw.sh stream4k --width 3840 --height 2160 --batch 64
echo "=== SQ counters, stream 1080p"; MI355_PIPELINE=0 bash profiles/pmc_sq.sh r04_stream
echo "=== SQ counters, pair mode 1080p"; MI355_PIPELINE=0 bash profiles/pmc_sq.sh r04_pair1080 --apart --batch 128
echo "=== SQ counters, pair mode 4K"; MI355_PIPELINE=0 bash profiles/pmc_sq.sh r04_pair4k --apart --width 3840 --height 2160 --batch 64
echo "=== SQ counters, S0"; MI355_PIPELINE=0 bash profiles/pmc_sq.sh r04_s0 --regime s0 --batch 32
echo "=== regimes"; python tools/bench_regimes.py
} > gpurun_out/r04prof/log.txt 2>&1
tail -5 gpurun_out/r04prof/log.txt
