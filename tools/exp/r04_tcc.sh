#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r04tcc
{
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' | cut -c1-3000
echo
for v in old p3; do
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  MI355_PIPELINE=0 LD_LIBRARY_PATH=build/ab/$v timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/r04tcc/$v -- tools/diffbench --steps 3 --warmup 1 > gpurun_out/r04tcc/$v.log 2>&1 || echo "failed $v $grp: $(tail -2 gpurun_out/r04tcc/$v.log)"
done
echo "== $v"
python3 - gpurun_out/r04tcc/$v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "mi355" in r["Kernel_Name"] and "probe" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
        print(f"   {c:28s} n={len(big)} avg={sum(big)/len(big):.4g}")
PY
done
} > gpurun_out/r04tcc/log.txt 2>&1
cat gpurun_out/r04tcc/log.txt | tail -60
