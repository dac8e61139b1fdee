#!/usr/bin/env python3
"""Do the fill and the walker launch of the walk service overlap?  Run under rocprofv3 --kernel-trace and read the trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ws -o ws -- python3 tools/walk_service_trace.py
    python3 tools/walk_service_trace.py --read gpurun_out/ws
"""
import csv
import glob
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def run():
    import numpy as np
    from caretta_amd import dynamic_time_warping as dtw
    rng = np.random.default_rng(1)
    n, count = 300, 8128
    base = rng.uniform(size=(64, n, n)) ** 3
    idx = np.arange(n)
    batch = dtw.ExplicitBatch([(idx, idx, base[k % 64]) for k in range(count)])
    for _ in range(3):
        batch.smith_waterman(0.0)
    for _ in range(3):
        batch.dtw_align(1.0, 0.01, want_alignments=True)
    batch.close()


def read(folder):
    rows = []
    for f in glob.glob(f"{folder}/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = None
    for r in rows:
        name = r["Kernel_Name"]
        if not any(k in name for k in ("k_sw_trace_rows", "walk_service", "k_explicit_stream")):
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        t0 = t0 or s
        print(f"{name[:60]:60s} start {(s - t0) / 1e3:10.1f} us  end {(e - t0) / 1e3:10.1f} us  ({(e - s) / 1e3:8.1f} us)  queue {r.get('Queue_Id')} grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        read(sys.argv[2])
    else:
        run()
