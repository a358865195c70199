#!/bin/bash
# r05au: the bench lines of tools/exp/r05_final.sh once more, with the final bench.py (secondary lines warmed, steady_state)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05au; mkdir -p $O
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "default rc=$?"
MI355_PIPELINE=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-host-path --no-config5 > $O/bench_sequential.json 2>/dev/null; echo "sequential rc=$?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --width 3840 --height 2160 --batch 64 --no-cpu --no-host-path --no-filters --no-pair > $O/bench_4k.json 2>/dev/null; echo "4k rc=$?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --width 3840 --height 2160 --batch 64 --shard roundrobin --no-cpu --no-host-path --no-filters --no-pair > $O/bench_4k_roundrobin.json 2>/dev/null; echo "4k rr rc=$?"
timeout -k 10 400 python bench.py --gpus 4 --rehearse-on-one-gpu --batch 64 --steps 5 --warmup 2 --gather-every-steps 6 --no-cpu > $O/bench_rehearsal4.json 2> $O/bench_rehearsal4.err; echo "rehearsal rc=$?"
