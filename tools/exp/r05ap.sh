#!/bin/bash
# r05ap: the headline's 20-step window against the time the chip has been under this load (warm-up steps of 0.47 ms each)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ap; mkdir -p $O; : > $O/log.txt
one() { timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-cpu --no-pair --no-filters --no-host-path --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2:', d['value'], d['ms_per_step'], d['roofline']['frac'], [k['avg_us'] for k in d['roofline']['kernels']], d['board']['rocm_smi_after'].get('Current Socket Graphics Package Power (W)'))" >> $O/log.txt; }
for rep in 1 2; do for w in 5 25 50 100 200 400 1000 2000 4000; do one 20 $w; done; one 1000 10; one 4000 10; done
cat $O/log.txt
