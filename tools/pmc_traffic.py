#!/usr/bin/env python3
"""profiles/pmc_traffic.json (read by bench.py -> roofline.traffic) from a tools/pmc_summary.py summary.

    python tools/pmc_traffic.py gpurun_out/<tag>/summary.json profiles/pmc_traffic.json [workload] [n_gpus]

HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE x 1 (KiB counters; gfx950 counts a 128-byte read request as 64 bytes,
MI355X_MICROARCH.md), each from its own --pmc pass.
"""
import json
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "headline"
    gpus = sys.argv[4] if len(sys.argv) > 4 else "1"
    summary = json.load(open(src))
    try:
        out = json.load(open(dst))
    except (OSError, ValueError):
        out = {}
    for name, c in summary.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        kernel = "k_seed" if "k_seed" in name else "k_align" if "k_align" in name else None
        if kernel is None:
            continue
        fetch, write = c["FETCH_SIZE"] * 1024.0 * 2.0, c["WRITE_SIZE"] * 1024.0
        out[f"{workload}:{gpus}:{kernel}"] = {
            "hbm_bytes_per_launch": fetch + write,
            "fetch_bytes_corrected": fetch,
            "write_bytes": write,
            "raw_FETCH_SIZE_KiB": c["FETCH_SIZE"],
            "raw_WRITE_SIZE_KiB": c["WRITE_SIZE"],
            "kernel": name,
            "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B), WRITE_SIZE x1",
            "source": f"{src} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc_profile.sh)",
        }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
