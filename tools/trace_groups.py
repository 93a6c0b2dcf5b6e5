#!/usr/bin/env python3
"""Average duration per (kernel name, grid size) of a rocprofv3 --kernel-trace CSV (developer tool):
python tools/trace_groups.py <kernel_trace.csv> [name-substring]"""
import csv
import sys
from collections import defaultdict

sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:70], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, wgs), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{name:70s} wgs={wgs:6d} n={len(v):4d} median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}")
