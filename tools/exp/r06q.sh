#!/bin/bash
# r06q: the sustained figure of round 6's library: 40 000 back-to-back batches (10.24 M frames, ~18 s) twice in a row, as r05as.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06q; mkdir -p $O; : > $O/summary.txt
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 40000 --warmup 10 --preheat-s 0 --steady-steps 0 --no-cpu --no-pair --no-filters --no-host-path --no-config5 > $O/long$i.json 2> $O/long$i.err
  python3 -c "
import json
x=json.loads(open('$O/long$i.json').read().strip().splitlines()[-1]); r=x['roofline']
print('run $i:', x['value'], 'frames/s', x['ms_per_step'], 'ms', 'frac', r['frac'], 'actual', r.get('frac_actual'), 'of achievable', r.get('frac_of_achievable'), 'board read', x['board']['hbm_stream_read_gbps'], 'power', (x['board'].get('rocm_smi_after') or {}).get('Current Socket Graphics Package Power (W)'))" | tee -a $O/summary.txt
done
