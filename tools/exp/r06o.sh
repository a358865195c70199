#!/bin/bash
# r06o: 4K round-robin pairs as the whole job, one rank: without a launcher, under the launcher without a group (--gather none), under
# the launcher with the RCCL group -- what in the process makes pair mode 20 % slower there?
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06o; mkdir -p $O; : > $O/summary.txt
A="--gpus 1 --steps 20 --warmup 5 --width 3840 --height 2160 --batch 64 --shard roundrobin --no-cpu --no-pair --no-filters --no-host-path --no-config5 --preheat-s 1 --steady-steps 200"
show() { python3 -c "
import json
x=json.loads(open('$2').read().strip().splitlines()[-1])
print('$1', 'value', x['value'], 'ms', x['ms_per_step'], 'frac', x['roofline']['frac'], 'kernels', [k['avg_us'] for k in x['roofline']['kernels']], 'steady', x['steady_state']['ms_per_step'])" | tee -a $O/summary.txt; }
timeout -k 10 300 python bench.py $A > $O/plain.json 2> $O/plain.err; show plain $O/plain.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29571 bench.py $A --gather none > $O/l_none.json 2> $O/l_none.err; show launcher_gather_none $O/l_none.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29572 bench.py $A > $O/l_after.json 2> $O/l_after.err; show launcher_gather_after $O/l_after.json
OMP_NUM_THREADS=1 timeout -k 10 300 python bench.py $A > $O/plain_omp1.json 2> $O/plain_omp1.err; show plain_OMP1 $O/plain_omp1.json
