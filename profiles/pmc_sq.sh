#!/bin/bash
# SQ counter passes on the C++ harness (one rocprofv3 --pmc run per counter group).
set -u
TAG=${1:-sq}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
DB="tools/diffbench --steps 3 --warmup 1 $@"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_VALU_MFMA_BUSY_CYCLES" ; do
  i=$((i+1))
  # (profiles/pmc_pass.sh: ends the pass at once when rocprofv3 refuses the counter set instead of waiting for the timeout)
  bash profiles/pmc_pass.sh 200 $OUT/g$i.log rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- $DB || echo "group $i failed: $(tail -2 $OUT/g$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "mi355" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v)} avg={sum(v)/len(v):.4g}")
# The clock a kernel ran at: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / 8 / the launch's duration in the same pass's
# kernel trace.  Only for launches of 100 us and more (the counter's window is wider than a short kernel).
import os
cc = sorted(glob.glob(out + "/g3/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
kt = sorted(glob.glob(out + "/g3/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
if cc and kt:
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[-1]))}
    ghz = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[-1])):
        d = dur.get(r["Dispatch_Id"], 0)
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "mi355" in r["Kernel_Name"] and d >= 100000:
            ghz[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]) / 8.0 / d)
    print("shader clock during the kernel (GHz): GRBM_GUI_ACTIVE / 8 XCDs / duration, launches >= 100 us")
    for k, v in ghz.items():
        v.sort()
        print(f"   {k:42s} n={len(v)} min={v[0]:.3f} median={v[len(v)//2]:.3f} max={v[-1]:.3f}")
PY
