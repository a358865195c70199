#!/bin/bash
# Round 6's evidence run on the round's final library: suites first, then everything that lands in profiles/r06_*.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r06final; mkdir -p $O
say() { echo "$@" >> $O/log.txt; }
: > $O/log.txt
say "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -3 $O/suite.txt >> $O/log.txt
say "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $O/log.txt
say "== soak 3000"; timeout -k 10 500 python tests/soak.py 3000 > $O/soak.txt 2>&1; tail -2 $O/soak.txt >> $O/log.txt
say "== soak_chain 60"; timeout -k 10 400 python tests/soak_chain.py 60 9 > $O/soak_chain.txt 2>&1; tail -2 $O/soak_chain.txt >> $O/log.txt
say "== compat_pipe first frame"; for i in 1 2; do timeout -k 10 60 tools/compat_pipe 1920 1080 12 2>&1 | tail -1 >> $O/log.txt; done
say "== bench.py, the driver's command"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; say "rc=$?"
say "== bench.py --preheat-s 0 (rounds 1-5's window)"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 --preheat-s 0 --no-cpu --no-host-path --no-config5 --no-pair --no-filters > $O/bench_no_preheat.json 2>/dev/null; say "rc=$?"
say "== bench.py sequential"; MI355_PIPELINE=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-host-path --no-config5 > $O/bench_sequential.json 2>/dev/null; say "rc=$?"
say "== bench.py 4K stream"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 --width 3840 --height 2160 --batch 64 --no-cpu --no-host-path --no-filters --no-pair > $O/bench_4k.json 2>/dev/null; say "rc=$?"
say "== bench.py 4K round-robin pairs"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 --width 3840 --height 2160 --batch 64 --shard roundrobin --no-cpu --no-host-path --no-filters --no-pair > $O/bench_4k_roundrobin.json 2>/dev/null; say "rc=$?"
say "== bench.py rehearsal, 3 ranks on the one GPU (1080p 64-frame batches; config5: 4K, 16-frame shards)"; timeout -k 10 500 python bench.py --gpus 3 --rehearse-on-one-gpu --batch 64 --steps 5 --warmup 2 --gather-every-steps 4 --no-cpu --config5-size 3840 2160 16 --config5-steps 3 --steady-steps 100 --preheat-s 0.2 > $O/bench_rehearsal3.json 2> $O/bench_rehearsal3.err; say "rc=$?"
say "== bench.py under the launcher, 1 rank, real RCCL (the config5 object at full size)"; timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-pair --no-filters --no-host-path > $O/bench_launcher1.json 2> $O/bench_launcher1.err; say "rc=$?"
say "== run_profile.sh r06"; bash profiles/run_profile.sh r06 5 > $O/run_profile.txt 2>&1; tail -25 $O/run_profile.txt >> $O/log.txt
say "== filters"; bash profiles/run_profile_filters.sh r06 > $O/run_profile_filters.txt 2>&1; tail -25 $O/run_profile_filters.txt >> $O/log.txt
say "== pair 1080p apart"; bash profiles/pmc_fw.sh pair1080apart --apart --batch 128 > $O/pmc_pair1080apart.txt 2>&1
say "== pair 4K apart"; bash profiles/pmc_fw.sh pair4kapart --apart --width 3840 --height 2160 --batch 64 > $O/pmc_pair4kapart.txt 2>&1
say "== S0"; bash profiles/pmc_fw.sh s0 --regime s0 --batch 32 --lib-alloc > $O/pmc_s0.txt 2>&1
say "== SQ stream"; MI355_PIPELINE=0 bash profiles/pmc_sq.sh r06_stream > $O/sq_stream.txt 2>&1
say "== done"
tail -70 $O/log.txt
