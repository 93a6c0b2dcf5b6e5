#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; *_counter_collection.csv) of the same
bench.py command into per-kernel HBM-side bytes per launch.

  python tools/pmc_traffic.py FETCH.csv WRITE.csv OUT.json

Units and the gfx950 correction follow MI355X_MICROARCH.md: both counters are KiB; FETCH_SIZE
under-reports 16-byte-per-lane reads by exactly 2x, so traffic = (2*FETCH + WRITE) * 1024 bytes."""
import csv
import json
import sys
from collections import defaultdict


def fold(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]] += 1
    return tot, n


def main():
    import datetime
    import os
    import subprocess
    fetch, nf = fold(sys.argv[1], "FETCH_SIZE")
    write, nw = fold(sys.argv[2], "WRITE_SIZE")
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 "
                   "--warmup 1 --no-cpu-baseline --no-profile-events`; counters are KiB; per MI355X_MICROARCH.md "
                   "FETCH_SIZE under-reports wide (16 B/lane) reads by exactly 2x on gfx950, so traffic = "
                   "(2*FETCH + WRITE)*1024 bytes per launch", "kernels": {}}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:   # the tree the passes were folded in (the measured tree is this commit or a descendant's parent: see git log)
        out["_commit"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                        timeout=10).stdout.strip() or None
    except Exception:
        out["_commit"] = None
    out["_date"] = datetime.datetime.utcfromtimestamp(os.path.getmtime(sys.argv[1])).strftime("%Y-%m-%d %H:%M UTC")
    for k in fetch:
        if k not in write or nf[k] != nw[k]:
            continue
        f, w = fetch[k] / nf[k], write[k] / nw[k]
        out["kernels"][k] = {"launches": nf[k], "fetch_kib_per_launch": round(f, 1), "write_kib_per_launch": round(w, 1),
                             "traffic_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
