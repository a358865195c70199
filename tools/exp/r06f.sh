#!/bin/bash
# r06f: mi355_alloc_outputs with the expansion itself as the probe: eight fresh processes of the S0 regime with the output
# arrays from the library, eight with plain hipMalloc; bench.py's regimes object.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06f; mkdir -p $O; : > $O/summary.txt
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 100 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --lib-alloc > $O/lib$i.json 2> $O/lib$i.err || echo "lib $i failed" | tee -a $O/summary.txt
  timeout -k 10 100 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 > $O/plain$i.json 2> $O/plain$i.err || echo "plain $i failed" | tee -a $O/summary.txt
  echo "process $i: lib-alloc $(grep -o '"kernels_us": [^]]*]' $O/lib$i.json) ($(cat $O/lib$i.err | tr '\n' ' '))  plain $(grep -o '"kernels_us": [^]]*]' $O/plain$i.json)" | tee -a $O/summary.txt
done
for i in 1 2; do
  timeout -k 10 100 tools/diffbench --regime flip --batch 32 --steps 10 --warmup 30 --lib-alloc > $O/flip_lib$i.json 2> $O/flip_lib$i.err
  timeout -k 10 100 tools/diffbench --regime flip --batch 32 --steps 10 --warmup 30 > $O/flip_plain$i.json 2>/dev/null
  echo "P = N $i: lib-alloc $(grep -o '"kernels_us": [^]]*]' $O/flip_lib$i.json) ($(cat $O/flip_lib$i.err | tr '\n' ' '))  plain $(grep -o '"kernels_us": [^]]*]' $O/flip_plain$i.json)" | tee -a $O/summary.txt
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-host-path --no-config5 --steady-steps 0 --no-filters > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
python3 - $O/bench.json <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["regimes"]
    print("regimes", {k: (v.get("frac"), v.get("kernels_us")) for k, v in r.items() if k != "outputs"}, r.get("outputs"), r["S0_refrand_pairs"].get("plain_allocation"))
except Exception as e:
    print("bench line unreadable:", e)
PY
