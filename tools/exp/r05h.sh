#!/bin/bash
# r05h: the instruction budget of the two kernels, phase by phase: SQ_INSTS_VALU / SALU / LDS / VMEM of laboratory builds
# that stop after a phase (csrc/lab.h), one batch after the other (MI355_PIPELINE=0: one launch per kernel and batch)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp MI355_PIPELINE=0
O=gpurun_out/r05h; mkdir -p $O
for v in new xa9 xa1 xa2 xa3 pa1 pa2 pa3; do
  LD_LIBRARY_PATH=build/ab/$v timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/$v -- tools/diffbench --steps 3 --warmup 1 > $O/$v.log 2>&1 || echo "$v failed: $(tail -2 $O/$v.log)"
  echo "done $v" >> $O/progress.txt
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in ("new", "xa9", "xa1", "xa2", "xa3", "pa1", "pa2", "pa3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in glob.glob(f"{out}/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "mi355" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][-34:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for p in glob.glob(f"{out}/{v}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "mi355" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0][-34:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, d in acc.items():
        if "scan" in k or "probe" in k: continue
        us = sorted(dur[k])[len(dur[k]) // 2] / 1e3 if dur[k] else 0
        print(f"{v:5s} {k:36s} us={us:7.1f} " + " ".join(f"{c[3:]}={sum(x)/len(x):.4g}" for c, x in sorted(d.items())))
PY
