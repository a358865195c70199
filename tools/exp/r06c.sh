#!/bin/bash
# r06c: (1) config 3 in one read of the colour frames: parity tests, then the A/B in bench.py's config3 object;
# (2) the dense expansion's value array put together from scattered physical pieces (HIP virtual-memory calls), probes with the
# items dealt out of order / one array only, and per-instance L2 counters of fast and slow arrays in one process.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06c; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_filters_gpu.py -m gpu -x -q -k "config3 or binarize" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -15 $O/pytest.log | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-pair --no-cpu --no-host-path --no-config5 --steady-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
python3 - $O/bench.json <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("config3", json.dumps(d["config3"])[:900])
except Exception as e:
    print("bench line unreadable:", e)
PY
DIFFBENCH_PLACE_NO_SWEEPS=1 timeout -k 10 300 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --place 14 > $O/place1.log 2>&1 || echo "place 1 failed" | tee -a $O/summary.txt
grep -h "^place\|^probe2" $O/place1.log | cut -c1-150 | tee -a $O/summary.txt
DIFFBENCH_PLACE_NO_SWEEPS=1 timeout -k 5 200 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL --kernel-trace --output-format json -d $O/pmc -- tools/diffbench --regime s0 --batch 32 --steps 3 --warmup 2 --place 14 > $O/pmc.log 2>&1 || echo "pmc failed: $(grep -m1 -i 'error\|exceeds' $O/pmc.log | cut -c1-160)" | tee -a $O/summary.txt
find $O -name "*.json" -path "*pmc*" -size +1k | while read f; do gzip -9 "$f"; done
du -sh $O | tee -a $O/summary.txt
