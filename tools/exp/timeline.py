#!/usr/bin/env python3
"""Prints the kernels of a rocprofv3 --kernel-trace CSV as a timeline (us relative to the first pack kernel of the window)."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "mi355" not in n or "clock" in n:
        continue
    short = "pack" if "diff_pack" in n else "scan" if "scan" in n else "expand" if "expand" in n else n[:20]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", "")))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = rows[skip:skip + 18]
t0 = rows[0][0]
for s, e, n, st in rows:
    print(f"{n:7s} stream {st:>3s}  start {(s - t0) / 1e3:8.1f}  end {(e - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}")
