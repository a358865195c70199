#!/bin/bash
# r05v: bench.py with the exchange behind the timed steps (default) -- rehearsal tests, 4 ranks at 1080p both ways, N = 1
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05v; mkdir -p $O
echo "== rehearsal tests" > $O/log.txt; timeout -k 10 500 python -m pytest tests/test_rehearsal_gpu.py -x -q --timeout 300 > $O/t.txt 2>&1; tail -3 $O/t.txt >> $O/log.txt
for g in after last; do
  echo "== 4 ranks, --gather $g" >> $O/log.txt
  timeout -k 10 500 python bench.py --gpus 4 --rehearse-on-one-gpu --batch 64 --steps 20 --warmup 5 --gather $g --gather-every-steps 3 --no-cpu > $O/r4_$g.json 2> $O/r4_$g.err; echo "rc=$?" >> $O/log.txt
  python3 -c "
import json,sys
d=json.loads(open('$O/r4_$g.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','value_with_final_gather','final_gather_ms','ms_per_step','ranks_seen','gather_ms','gather_bytes','gather_verified')}, d['config']['gather'][:40])" >> $O/log.txt 2>&1
done
echo "== N=1 default" >> $O/log.txt; timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-pair --no-filters --no-host-path --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['gather'], d['ranks_seen'], d['gather_ms'])" >> $O/log.txt 2>&1
cat $O/log.txt
