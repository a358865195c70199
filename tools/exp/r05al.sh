#!/bin/bash
# r05al: one more default bench line on whatever box this call gets (the spread of the final library over the pool)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05al; mkdir -p $O
T=$(date -u +%H%M%S)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_$T.json 2> $O/bench_$T.err; echo "rc=$?"
python3 - $O/bench_$T.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["regimes"]
print(json.dumps({"value": d["value"], "ms": d["ms_per_step"], "frac": d["roofline"]["frac"], "traffic": d["roofline"]["traffic"],
                  "two_streams": d["two_streams_one_gpu"]["frac"], "pair": d["pair_mode"]["frac"], "S0": r["S0_refrand_pairs"]["frac"],
                  "S0_kernels_us": r["S0_refrand_pairs"]["kernels_us"], "PeqN": r["P_eq_N_pairs"]["frac"], "Peq0": r["P_eq_0_pairs"]["frac"],
                  "config3": [d["config3"]["us_per_frame"], d["config3"]["sequential_us_per_frame"]],
                  "config4": [d["config4"]["us_per_frame"], d["config4"]["frac"]], "median": d["median5x5"]["us_per_frame"],
                  "config5_per_gpu": d["config5_per_gpu"]["frac"], "cpu": [d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"]],
                  "parity": all(d["parity"].values()), "read_gbps": d["board"]["hbm_stream_read_gbps"]}))
PY
