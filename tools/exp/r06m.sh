#!/bin/bash
# r06m: MI355_FLAG_OWN_QUEUES in the product: bench.py under the launcher (it sets the flag) and without (it does not), the
# new test, the rehearsal tests.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06m; mkdir -p $O; : > $O/summary.txt
timeout -k 10 400 python -m pytest tests/test_diff_pack_gpu.py tests/test_rehearsal_gpu.py tests/test_group_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
A="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-pair --no-filters --no-host-path --preheat-s 1 --steady-steps 200"
run() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    x = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    c5 = x.get("config5") or x.get("config5_per_gpu") or {}
    print(sys.argv[1], "value", x["value"], "ms", x["ms_per_step"], "kernels", [k["avg_us"] for k in x["roofline"]["kernels"]], "steady", x["steady_state"]["ms_per_step"], "| config5 frac", c5.get("frac"), c5.get("value") or c5.get("frames_per_s"), "flags", x["config"].get("core_flags", "")[:24])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
}
for i in 1 2; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2955$i bench.py $A > $O/launcher_$i.json 2> $O/launcher_$i.err
  run "launcher $i" $O/launcher_$i.json | tee -a $O/summary.txt
  timeout -k 10 300 python bench.py $A > $O/plain_$i.json 2> $O/plain_$i.err
  run "plain    $i" $O/plain_$i.json | tee -a $O/summary.txt
done
