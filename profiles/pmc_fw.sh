#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate rocprofv3 --pmc runs, MI355X_MICROARCH.md) on the C++ harness for any
# diffbench configuration:  bash profiles/pmc_fw.sh TAG [diffbench args...]   e.g.  pmc_fw.sh pair1080 --pairs --batch 128
# Prints, per kernel of the library, the bytes per launch: read = 2 x FETCH_SIZE (gfx950 tallies a 128-B request as 64 B),
# write = WRITE_SIZE; raw CSVs under gpurun_out/pmc_fw_TAG/.
set -u
TAG=${1:-fw}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=$ROOT/gpurun_out/pmc_fw_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
DB="tools/diffbench --steps 3 --warmup 1 $@"
for ctr in FETCH_SIZE WRITE_SIZE; do
  MI355_PIPELINE=0 bash profiles/pmc_pass.sh 200 $OUT/$ctr.log rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -- $DB || echo "$ctr pass failed: $(tail -2 $OUT/$ctr.log)"
done
python3 - "$OUT" "$TAG" "$*" <<'PY'
import csv, glob, sys, collections, json
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "mi355" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                acc[r["Kernel_Name"].split("(")[0]][ctr].append(float(r["Counter_Value"]))
    for p in glob.glob(f"{out}/{ctr}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "mi355" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {"tag": tag, "diffbench_args": args, "kernels": {}}
tot = 0
for k, d in acc.items():
    if "probe" in k:
        continue
    f = d.get("FETCH_SIZE", [0]); w = d.get("WRITE_SIZE", [0])
    big_f = [x for x in f if x > 0.5 * max(f)] if max(f) > 0 else f
    big_w = [x for x in w if x > 0.5 * max(w)] if max(w) > 0 else w
    rd = 2 * sum(big_f) / len(big_f) * 1024; wr = sum(big_w) / len(big_w) * 1024
    us = sorted(dur[k])[len(dur[k]) // 2] / 1e3 if dur[k] else None
    res["kernels"][k] = {"read_bytes": int(rd), "write_bytes": int(wr), "median_us_under_profiler": us,
                         "tbps": round((rd + wr) / (us * 1e-6) / 1e12, 2) if us else None}
    tot += rd + wr
res["hbm_bytes_per_launch"] = int(tot)
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY
