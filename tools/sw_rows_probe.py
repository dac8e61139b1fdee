#!/usr/bin/env python3
"""smith_waterman with gap 0 and its traceback over the 8 128 x 300 x 300 list (k_sw_trace_rows): what the fill alone costs
(CARETTA_SW_ROWS_NOWALK), built for 3 / 4 / 5 waves per SIMD (CARETTA_SW_ROWS_WAVES), and the path it replaced
(CARETTA_NO_SW_ROWS).  Measurement tool: the switched variants' results are not checked (NOWALK's are wrong by design).

    python tools/sw_rows_probe.py [COUNT [N]]
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from caretta_amd import dynamic_time_warping as dtw, engine  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 8128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    rng = np.random.default_rng(1)
    base = rng.uniform(size=(64, n, n)) ** 3
    idx = np.arange(n)
    batch = dtw.ExplicitBatch([(idx, idx, base[k % 64]) for k in range(count)])
    nbytes = 8.25 * batch.cells + 24.0 * count * 2 * n + 8.0 * count
    variants = [("row sweep + walk (product)", {}), ("row sweep, fill only", {"CARETTA_SW_ROWS_NOWALK": "1"}),
                ("row sweep + walk, 5 waves/SIMD build", {"CARETTA_SW_ROWS_WAVES": "5"}), ("row sweep + walk, 3 waves/SIMD build", {"CARETTA_SW_ROWS_WAVES": "3"}),
                ("fill only, 5 waves/SIMD build", {"CARETTA_SW_ROWS_WAVES": "5", "CARETTA_SW_ROWS_NOWALK": "1"}),
                ("skewed sweep + walk (the gap != 0 kernel)", {"CARETTA_NO_SW_ROWS": "1"})]
    for name, env in variants:
        for k in list(os.environ):
            if k.startswith("CARETTA_SW_ROWS") or k == "CARETTA_NO_SW_ROWS":
                del os.environ[k]
        os.environ.update(env)
        engine.reload_config()
        batch.smith_waterman(0.0)
        ms = []
        for _ in range(3):
            batch.smith_waterman(0.0)
            ms.append(batch.last_kernel_ms())
        print(f"{name:45s}: {min(ms):.3f} ms (median {sorted(ms)[1]:.3f}) -> {nbytes / min(ms) / 1e6:.0f} GB/s = {nbytes / min(ms) / 1e6 / 8000:.3f} of peak", flush=True)
    nb = 8.5 * batch.cells + 24.0 * count * 2 * n + 8.0 * count
    batch.dtw_align(1.0, 0.01, want_alignments=True)
    ms = []
    for _ in range(3):
        batch.dtw_align(1.0, 0.01, want_alignments=True)
        ms.append(batch.last_kernel_ms())
    print(f"{'dtw_align with its traceback':45s}: {min(ms):.3f} ms (median {sorted(ms)[1]:.3f}) -> {nb / min(ms) / 1e6:.0f} GB/s = {nb / min(ms) / 1e6 / 8000:.3f} of peak", flush=True)
    batch.close()


if __name__ == "__main__":
    main()
