#!/usr/bin/env python3
"""Condenses the rocprofv3 output of profiles/run_profile_filters.sh into committed summaries:
  profiles/<tag>_filters_kernel_stats.csv : rocprofv3 --stats rows of the library's kernels
  profiles/<tag>_filters_pmc.json         : per kernel, per launch of a 96-frame 1080p batch: average duration
        (kernel trace), FETCH_SIZE / WRITE_SIZE (separate --pmc passes; KiB -> bytes, reads doubled as
        MI355X_MICROARCH.md prescribes for gfx950), bytes moved per frame and the fraction of 8 TB/s
  profiles/<tag>_filters_lines.jsonl      : the harness's own lines of the same command (no profiler attached)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
B = 96


def find(sub, pat):
    # gpurun merges every call's output into gpurun_out/: a directory may hold the files of earlier calls (other process
    # numbers) beside this one's -- the newest is this call's
    r = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return max(r, key=os.path.getmtime) if r else None


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def short(name):
    return name.split("(")[0].split("::")[-1]


stats = find("trace", "*kernel_stats.csv")
if stats:
    rs = rows(stats)
    keep = [r for r in rs if "mi355" in r.get("Name", "")]
    with open(os.path.join(here, f"{tag}_filters_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rs[0].keys()))
        w.writeheader()
        w.writerows(keep)

dur = defaultdict(list)
trace = find("trace", "*kernel_trace.csv")
if trace:
    for r in rows(trace):
        if "mi355" in r["Kernel_Name"]:
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

ctr = defaultdict(dict)
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    p = find(sub, "*counter_collection.csv")
    if not p:
        continue
    vals = defaultdict(list)
    for r in rows(p):
        if r.get("Counter_Name") == name and "mi355" in r["Kernel_Name"]:
            vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        big = [x for x in v if x > 0.5 * max(v)] if max(v) > 0 else v
        ctr[k][name] = sum(big) / len(big) * 1024.0

out = {"tag": tag, "batch": B, "width": 1920, "height": 1080,
       "note": "per launch on a 96-frame 1080p batch; read_bytes = 2 x FETCH_SIZE (gfx950 tallies a 128-B read request "
               "as 64 B, MI355X_MICROARCH.md), write_bytes = WRITE_SIZE; launches of the diff path inside the chains "
               "(k_diff_pack, k_scan_groups, k_expand) appear with their 96-frame figures",
       "kernels": {}}
for k, v in sorted(dur.items()):
    big = [x for x in v if x > 0.5 * max(v)]
    avg_us = sum(big) / len(big) / 1e3
    e = {"launches": len(v), "avg_us": round(avg_us, 2), "us_per_frame": round(avg_us / B, 3)}
    c = ctr.get(k, {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = 2 * c["FETCH_SIZE"], c["WRITE_SIZE"]
        e.update({"read_bytes": int(rd), "write_bytes": int(wr), "bytes_per_frame": int((rd + wr) / B),
                  "moved_gbps": round((rd + wr) / (avg_us * 1e-6) / 1e9, 1),
                  "frac_of_8TBps": round((rd + wr) / (avg_us * 1e-6) / 1e9 / 8000.0, 4)})
    out["kernels"][k] = e
json.dump(out, open(os.path.join(here, f"{tag}_filters_pmc.json"), "w"), indent=1)
lines = os.path.join(src, "lines.jsonl")
if os.path.exists(lines):
    shutil.copy(lines, os.path.join(here, f"{tag}_filters_lines.jsonl"))
print(json.dumps(out, indent=1)[:3000])
