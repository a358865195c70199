#!/bin/bash
# r06k: under the launcher (torch.distributed + RCCL in the process) the batches of one core no longer overlap (kernels 399 / 11 / 107 us
# one after the other, 462 k frames/s instead of 618 k): HIP streams share the process's hardware queues (GPU_MAX_HW_QUEUES, default 4).
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06k; mkdir -p $O; : > $O/summary.txt
A="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-pair --no-filters --no-host-path --no-config5 --preheat-s 1 --steady-steps 200"
run() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    x = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", x["value"], "ms", x["ms_per_step"], "kernels", [k["avg_us"] for k in x["roofline"]["kernels"]], "steady", x["steady_state"]["ms_per_step"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2953$((RANDOM % 10)) bench.py $A > $O/launcher_$q.json 2> $O/launcher_$q.err
  run "launcher GPU_MAX_HW_QUEUES=$q" $O/launcher_$q.json | tee -a $O/summary.txt
  timeout -k 10 300 python bench.py $A > $O/plain_$q.json 2> $O/plain_$q.err
  run "plain    GPU_MAX_HW_QUEUES=$q" $O/plain_$q.json | tee -a $O/summary.txt
done
