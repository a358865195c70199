#!/bin/bash
# r05j: the suite with the options test and the reworked inter-process stand-in; the 4-rank rehearsal at 1080p; bench with traffic
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
echo "== gpu suite" > $O/log.txt; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1; tail -3 $O/suite.txt >> $O/log.txt
echo "== rehearsal, 4 ranks, 1080p, 64-frame batches" >> $O/log.txt
timeout -k 10 500 python bench.py --gpus 4 --rehearse-on-one-gpu --batch 64 --steps 5 --warmup 2 --gather-every-steps 3 --no-cpu > $O/bench_rehearsal4.json 2> $O/bench_rehearsal4.err; echo "rc=$?" >> $O/log.txt
echo "== bench" >> $O/log.txt; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/log.txt
cat $O/log.txt; head -c 900 $O/bench_rehearsal4.json
