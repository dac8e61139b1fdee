#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run:  python tools/trace_gaps.py <dir> [name-filter]

Reads *kernel_trace.csv under <dir>, orders the dispatches by start time and prints, for the dispatches whose kernel name
contains the filter (default: every kernel), the summed kernel time, the summed gaps between consecutive kernels and the
largest gaps -- what a launch sequence queued without host round trips still loses between dependent launches."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for path in glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0].replace("cr::", "")))
rows.sort()
rows = [r for r in rows if flt in r[2]] if flt else rows
busy = sum(e - s for s, e, _ in rows)
gaps = [(rows[k + 1][0] - rows[k][1], rows[k][2], rows[k + 1][2]) for k in range(len(rows) - 1)]
small = [g for g in gaps if 0 <= g[0] < 200_000]                    # (larger: host work between two calls)
per = defaultdict(lambda: [0, 0])
for g, a, b in small:
    per[(a[:28], b[:28])][0] += g
    per[(a[:28], b[:28])][1] += 1
print(f"{len(rows)} dispatches, kernel time {busy / 1e6:.3f} ms, gaps below 0.2 ms: {sum(g[0] for g in small) / 1e6:.3f} ms in {len(small)} gaps "
      f"(mean {sum(g[0] for g in small) / max(len(small), 1) / 1e3:.2f} us)")
for (a, b), (tot, cnt) in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  {a:28s} -> {b:28s}: {cnt:5d} gaps, mean {tot / cnt / 1e3:6.2f} us, total {tot / 1e6:7.3f} ms")
