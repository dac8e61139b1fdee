#!/usr/bin/env python3
"""profiles/<round>/explicit_batch_pmc.json from the FETCH_SIZE / WRITE_SIZE passes tools/refresh_profiles.sh makes over
tools/explicit_batch_rate.py (rocprofv3 --pmc, one counter per pass, --kernel-trace only).

    python tools/pmc_explicit_summary.py gpurun_out/<round> [COUNT N] > profiles/<round>/explicit_batch_pmc.json

HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KiB counters; gfx950 tallies a 128-byte read request as 64 bytes,
MI355X_MICROARCH.md); algorithmic bytes = 8 x COUNT x N x N (every byte of S once).
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(root, counter):
    acc = defaultdict(list)
    for path in glob.glob(f"{root}/explicit_{counter}/**/*counter_collection.csv", recursive=True):
        per_dispatch, names = defaultdict(float), {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                per_dispatch[row["Dispatch_Id"]] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"].replace("void ", "").split("(")[0].replace("cr::", "")
        for did, v in per_dispatch.items():
            acc[names[did]].append(v)
    return acc


def main():
    root = sys.argv[1]
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 8128
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    alg = 8.0 * count * n * n
    fetch, write = per_kernel(root, "FETCH_SIZE"), per_kernel(root, "WRITE_SIZE")
    kernels = {}
    for k in sorted(fetch):
        if not k.startswith("k_"):
            continue
        f = sum(fetch[k]) / len(fetch[k])
        w = sum(write[k]) / len(write[k]) if write.get(k) else 0.0
        hbm = f * 1024 * 2 + w * 1024
        kernels[k] = {"FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "dispatches": len(fetch[k]), "hbm_bytes_per_launch": hbm,
                      "matrix_bytes": alg, "traffic_over_matrix_bytes": hbm / alg}
    # per FUNCTION of the reference (tools/explicit_batch_rate.py): the kernels of its launch sequence against SURVEY 8(d)'s
    # explicit-mode bytes -- 8 n m + decisions (n m / 2 dtw_align, n m / 4 smith_waterman, none for a score) + 24 (n + m) + 8
    cells, small = float(count) * n * n, 24.0 * count * 2 * n + 8.0 * count
    small_score = 8.0 * count * 2 * n + 8.0 * count          # (a score alone writes no alignment rows)
    functions = {"smith_waterman_score gap 0 (row sweep)": (["k_sw_score_rows<"], 8.0 * cells + small_score),
                 "smith_waterman_score gap 0.1 (skewed sweep)": (["k_explicit_stream<1, 2, false>"], 8.0 * cells + small_score),
                 "dtw_align_score (skewed sweep, no decisions)": (["k_explicit_stream<1, 4, false>"], 8.0 * cells + small_score),
                 "dtw_align WITH traceback (4-bit decisions + walk)": (["k_explicit_stream<1, 4, true>", "k_dtw_trace_batch<"], 8.5 * cells + small),
                 "smith_waterman WITH traceback gap 0 (row sweep: 2-bit decisions + walk in one launch)": (["k_sw_trace_rows<"], 8.25 * cells + small),
                 "smith_waterman WITH traceback gap 0.1 (skewed sweep + walk, one launch)": (["k_explicit_sw_batch<"], 8.25 * cells + small)}
    ratios = {}
    for name, (prefixes, nbytes) in functions.items():
        hbm = sum(v["hbm_bytes_per_launch"] for k, v in kernels.items() if any(k.startswith(p) for p in prefixes))
        if hbm:
            ratios[name] = {"hbm_bytes": hbm, "algorithmic_bytes": nbytes, "traffic_over_algorithmic": hbm / nbytes}
    print(json.dumps({"kernels": kernels, "traffic_over_algorithmic": ratios,
                      "note": "FETCH_SIZE x 2 + WRITE_SIZE (KiB counters, separate --pmc passes; gfx950 counts a 128-byte read request as 64 bytes)"}, indent=1))


if __name__ == "__main__":
    main()
