#!/bin/bash
# r06a: first GPU call of round 6 -- the GPU suite on the new library (mi355_prepare, 33-bit launch tags, bench.py's config5 leg
# rehearsed with 2 and 3 processes), the driver's bench command, the C++ drop-in's first-frame latency, and WHERE the dense
# expansion's two output arrays have to lie (tools/diffbench --place; per-instance L2 counters of a fast and a slow draw).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
for i in 1 2; do timeout -k 10 60 tools/compat_pipe 1920 1080 12 2>&1 | tail -1 | tee -a $O/summary.txt; done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
python3 - $O/bench.json <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "actual", r.get("frac_actual"), "of_achievable", r.get("frac_of_achievable"))
    print("cold", d.get("cold_start_window"), "steady", {k: d["steady_state"][k] for k in ("ms_per_step", "frac", "frac_actual")})
    print("S0", d["regimes"]["S0_refrand_pairs"]["frac"], d["regimes"]["S0_refrand_pairs"]["kernels_us"], "PeqN", d["regimes"]["P_eq_N_pairs"]["frac"], "pair", d["pair_mode"]["frac"],
          "c3", d["config3"]["us_per_frame"], d["config3"]["sequential_us_per_frame"], "c4", d["config4"]["us_per_frame"], "c5", d["config5_per_gpu"]["frac"])
    print("parity", d["parity"])
except Exception as e:
    print("bench line unreadable:", e)
PY
for i in 1 2 3; do
  timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --place 4 > $O/place$i.log 2>&1 || echo "place $i failed" | tee -a $O/summary.txt
done
for i in 1 2 3; do
  timeout -k 5 60 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_EA0_WRREQ_64B --kernel-trace --output-format json csv -d $O/pmc$i -- tools/diffbench --regime s0 --batch 32 --steps 4 --warmup 2 > $O/pmc$i.log 2>&1 || echo "pmc $i failed: $(grep -m1 -i 'error\|exceeds' $O/pmc$i.log | cut -c1-160)" | tee -a $O/summary.txt
done
find $O -name "*.json" -path "*pmc*" -size +1k | while read f; do gzip -9 "$f"; done
du -sh $O | tee -a $O/summary.txt
