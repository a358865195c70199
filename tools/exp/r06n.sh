#!/bin/bash
# r06n: the config5 object under the launcher (1 rank, real RCCL) against config5_per_gpu without a launcher, same box.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06n; mkdir -p $O; : > $O/summary.txt
A="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-pair --no-filters --no-host-path --preheat-s 1 --steady-steps 200"
for i in 1 2; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2956$i bench.py $A > $O/launcher_$i.json 2> $O/launcher_$i.err
  python3 -c "
import json,sys
x=json.loads(open('$O/launcher_$i.json').read().strip().splitlines()[-1]); c=x['config5']
print('launcher $i value', x['value'], 'config5', c['value'], c['ms_per_step'], c['frac'], c['frac_per_rank'], c['final_gather_ms'], c['gather_gbps'], c['parity'], c['gather_verified'])" | tee -a $O/summary.txt
  timeout -k 10 300 python bench.py $A > $O/plain_$i.json 2> $O/plain_$i.err
  python3 -c "
import json,sys
x=json.loads(open('$O/plain_$i.json').read().strip().splitlines()[-1]); c=x['config5_per_gpu']
print('plain $i value', x['value'], 'config5_per_gpu', c['frames_per_s'], c['ms_per_launch'], c['frac'], c['frac_sequential'], c['kernels_us'])" | tee -a $O/summary.txt
done
