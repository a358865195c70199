#!/bin/bash
# r06l: the core's three streams created with stream priorities of their own (hardware queues are pooled per priority class), under
# the launcher and without: lowprio = all three at the least priority, mixprio = own normal / side greatest / second pack stream least.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06l; mkdir -p $O; : > $O/summary.txt
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
for v in product lowprio mixprio product lowprio; do
  if [ $v = product ]; then unset MI355DIFF_LIB; else export MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so; fi
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2954$((RANDOM % 10)) bench.py $A > $O/launcher_$v.json 2> $O/launcher_$v.err
  run "launcher $v" $O/launcher_$v.json | tee -a $O/summary.txt
  timeout -k 10 300 python bench.py $A > $O/plain_$v.json 2> $O/plain_$v.err
  run "plain    $v" $O/plain_$v.json | tee -a $O/summary.txt
done
