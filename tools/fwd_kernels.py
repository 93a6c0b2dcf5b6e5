#!/usr/bin/env python3
"""Per-kernel timeline of the LAST forward in a rocprofv3 --kernel-trace CSV of tools/trace_fwd.py:
python tools/fwd_kernels.py <kernel_trace.csv> [first-kernel-substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "im2col"
start = max(i for i, r in enumerate(rows) if first in r["Kernel_Name"])
seg = rows[start:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:100]}  wgs={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}")
print(len(seg), "launches,", (int(seg[-1]["End_Timestamp"]) - t0) / 1e3, "us")
