#!/bin/bash
# r06d: (1) six fresh processes of the S0 regime with the output arrays from mi355_alloc_outputs (and six with plain
# hipMalloc): the dense expansion's time in each; (2) config 3 both ways per kernel (rocprofv3 --kernel-trace --stats) and
# in bytes fetched / written (separate --pmc passes), tools/diffbench --filters; (3) bench.py's regimes object.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06d; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout -k 10 100 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --lib-alloc > $O/lib$i.json 2> $O/lib$i.err || echo "lib $i failed" | tee -a $O/summary.txt
  timeout -k 10 100 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 > $O/plain$i.json 2> $O/plain$i.err || echo "plain $i failed" | tee -a $O/summary.txt
  echo "process $i: lib-alloc $(grep -o '"kernels_us": [^]]*]' $O/lib$i.json) ($(cat $O/lib$i.err | tr '\n' ' '))  plain $(grep -o '"kernels_us": [^]]*]' $O/plain$i.json)" | tee -a $O/summary.txt
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- tools/diffbench --filters --batch 192 --steps 10 > $O/filters_trace.log 2>&1 || echo "trace failed" | tee -a $O/summary.txt
grep '"chain"' $O/filters_trace.log | cut -c1-200 | tee -a $O/summary.txt
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/filters_kernel_stats.csv && head -30 $f | cut -d, -f1-6 | cut -c1-200 | tee -a $O/summary.txt
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- tools/diffbench --filters --batch 192 --steps 2 > $O/pmc_f.log 2>&1 || echo "pmc fetch failed: $(grep -m1 -i 'error\|exceeds' $O/pmc_f.log | cut -c1-160)" | tee -a $O/summary.txt
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- tools/diffbench --filters --batch 192 --steps 2 > $O/pmc_w.log 2>&1 || echo "pmc write failed: $(grep -m1 -i 'error\|exceeds' $O/pmc_w.log | cut -c1-160)" | tee -a $O/summary.txt
python3 - $O <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for tag, name in (("pmc_f", "FETCH_SIZE"), ("pmc_w", "WRITE_SIZE")):
    cc = glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True)
    if not cc: print(tag, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] == name: acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(tag, name, "%-90s launches %3d  mean per launch %.1f KB (x 1024 B; FETCH_SIZE: double it for bytes)" % (k, len(v), sum(v) / len(v)))
PY
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
du -sh $O | tee -a $O/summary.txt
