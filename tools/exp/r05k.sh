#!/bin/bash
# r05k: the inter-process stand-in after the slot-identity fix: rehearsal tests, then 4 ranks at 1080p with a gather after every batch
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05k; mkdir -p $O
echo "== rehearsal tests" > $O/log.txt; timeout -k 10 500 python -m pytest tests/test_rehearsal_gpu.py -x -q --timeout 300 > $O/t.txt 2>&1; tail -2 $O/t.txt >> $O/log.txt
echo "== rehearsal, 4 ranks, 1080p, 64-frame batches" >> $O/log.txt
timeout -k 10 500 python bench.py --gpus 4 --rehearse-on-one-gpu --batch 64 --steps 5 --warmup 2 --gather-every-steps 6 --no-cpu > $O/bench_rehearsal4.json 2> $O/bench_rehearsal4.err; echo "rc=$?" >> $O/log.txt
echo "== rehearsal, 6 ranks, round-robin 4K, 16-frame batches" >> $O/log.txt
timeout -k 10 500 python bench.py --gpus 6 --rehearse-on-one-gpu --shard roundrobin --width 3840 --height 2160 --batch 16 --steps 4 --warmup 2 --gather-every-steps 4 --no-cpu > $O/bench_rehearsal6.json 2> $O/bench_rehearsal6.err; echo "rc=$?" >> $O/log.txt
cat $O/log.txt; head -c 1500 $O/bench_rehearsal4.json; echo; grep -o '"ranks_seen": [0-9]*, "gather_ms": [0-9.]*, "gather_bytes": [0-9]*' $O/bench_rehearsal6.json; grep -o '"gather_verified": [a-z]*' $O/bench_rehearsal*.json
