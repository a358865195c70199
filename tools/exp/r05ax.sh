#!/bin/bash
# r05ax: L2 / fabric counters of the dense expansion in processes that drew the fast lot and processes that drew the slow one
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05ax; mkdir -p $O
G1="TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_LEVEL"
G2="TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_BUSY TCC_CYCLE"
G3="TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL"
G4="TCC_HIT TCC_MISS TCC_WRITE TCC_READ"
# (at most four counters of the L2 per pass: eight were refused -- "exceeds the capabilities of the hardware" -- and the
# refused process then sat until its timeout)
for i in 1 2 3; do for g in 1 2 3 4; do
  eval grp=\$G$g
  timeout -k 5 40 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/r${i}g$g -- tools/diffbench --regime s0 --batch 32 --steps 6 --warmup 2 > $O/r${i}g$g.log 2>&1 || { echo "run $i group $g failed: $(grep -m1 -i "error\|exceeds" $O/r${i}g$g.log | cut -c1-160)"; }
done; done
python3 - $O <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
rows = []
for d in sorted(glob.glob(out + "/r*g*")):
    if not os.path.isdir(d): continue
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True); kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not cc or not kt: continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
    acc = collections.defaultdict(list); du = []
    for r in csv.DictReader(open(cc[0])):
        if "k_expand" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            du.append(dur.get(r["Dispatch_Id"], 0) / 1e3)
    if not du: continue
    du.sort()
    print(os.path.basename(d), "k_expand median %.1f us" % du[len(du)//2], " ".join("%s=%.4g" % (k, sum(v)/len(v)) for k, v in sorted(acc.items())))
PY
