#!/bin/bash
# r06p: two cores with MI355_FLAG_OWN_QUEUES in one process (bench.py under the launcher: the job's core and the config5 core): GPU_MAX_HW_QUEUES 8 against the default 4.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06p; mkdir -p $O; : > $O/summary.txt
A="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-pair --no-filters --no-host-path --preheat-s 1 --steady-steps 200"
for q in 8 default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2958$((RANDOM % 10)) bench.py $A > $O/launcher_$q.json 2> $O/launcher_$q.err
  python3 -c "
import json
x=json.loads(open('$O/launcher_$q.json').read().strip().splitlines()[-1]); c=x['config5']
print('launcher queues=$q value', x['value'], x['ms_per_step'], 'config5', c['value'], c['ms_per_step'], c['frac'])" | tee -a $O/summary.txt
  timeout -k 10 300 python bench.py $A > $O/plain_$q.json 2> $O/plain_$q.err
  python3 -c "
import json
x=json.loads(open('$O/plain_$q.json').read().strip().splitlines()[-1]); c=x['config5_per_gpu']
print('plain queues=$q value', x['value'], x['ms_per_step'], 'config5_per_gpu', c['frames_per_s'], c['ms_per_launch'], c['frac'])" | tee -a $O/summary.txt
done
