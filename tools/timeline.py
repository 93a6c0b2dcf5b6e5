#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace database (rocpd sqlite):
per-stream busy time, GPU-idle gaps, and the kernels in launch order with their stream.
usage: python tools/timeline.py results.db [step_index_from_end]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = list(db.execute("select name, stream_id, start, end from kernels order by start"))
    # a step starts at the first launch after the optimizer's last kernel
    opt = lambda n: "multi_tensor_apply" in n or "adam_kernel" in n   # torch.optim.Adam / optim.FusedAdam
    bounds = [i for i, r in enumerate(rows) if opt(r[0]) and (i + 1 == len(rows) or not opt(rows[i + 1][0]))]
    lo, hi = bounds[-back - 1] + 1, bounds[-back] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][2], max(r[3] for r in step)
    print(f"step: {len(step)} launches, {(t1 - t0) / 1e6:.3f} ms")
    streams = sorted({r[1] for r in step})
    for s in streams:
        busy = sum(r[3] - r[2] for r in step if r[1] == s)
        print(f"  stream {s}: busy {busy / 1e6:7.3f} ms, {sum(1 for r in step if r[1] == s)} launches")
    ev = sorted([(r[2], 1) for r in step] + [(r[3], -1) for r in step])
    depth, last, idle, both = 0, t0, 0, 0
    for t, d in ev:
        if depth == 0:
            idle += t - last
        if depth >= 2:
            both += t - last
        depth += d
        last = t
    print(f"  GPU idle (no kernel in flight) {idle / 1e6:.3f} ms; >=2 kernels in flight {both / 1e6:.3f} ms")
    if len(sys.argv) > 3:
        for r in step:
            print(f"{(r[2] - t0) / 1e3:9.1f} +{(r[3] - r[2]) / 1e3:8.1f} us  s{r[1]}  {r[0][:90]}")


if __name__ == "__main__":
    main()
