#!/usr/bin/env python3
"""Read rate of the batched explicit-score-matrix kernels: COUNT matrices of N x N doubles resident in HBM.

    python tools/explicit_batch_rate.py [COUNT [N]]        default 8128 x 300 x 300 = 5.85 GB (BASELINE config 3's pair count)

Prints the device time of one smith_waterman_score_batch launch (HIP events on the launch stream), the algorithmic
read rate (8 bytes per cell / that time) against the 8 TB/s HBM peak, and the same for dtw_align_batch (scores only).
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from caretta_amd import dynamic_time_warping as dtw  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 8128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(1)
base = rng.uniform(size=(64, n, n)) ** 3                 # 64 distinct matrices, tiled: the kernels still read every byte
idx = np.arange(n)
t0 = time.perf_counter()
batch = dtw.ExplicitBatch([(idx, idx, base[k % 64]) for k in range(count)])
t_up = time.perf_counter() - t0
nbytes = 8.0 * batch.cells
for name, fn in (("smith_waterman_score_batch (gap 0, row sweep)", lambda: batch.smith_waterman_scores(0.0)),
                 ("smith_waterman_score_batch (gap 0.1, skewed sweep)", lambda: batch.smith_waterman_scores(0.1)),
                 ("dtw_align_batch (scores only)", lambda: batch.dtw_align(1.0, 0.01, want_alignments=False))):
    out = fn()
    ms = []
    for _ in range(5):
        out = fn()
        ms.append(batch.last_kernel_ms())
    best = min(ms)
    print(f"{name}: {count} x {n} x {n}: {best:.3f} ms (median {sorted(ms)[2]:.3f}) -> {nbytes / best / 1e6:.0f} GB/s "
          f"= {nbytes / best / 1e6 / 8000:.3f} of the 8 TB/s peak; checksum {float(np.sum(out)):.6f}")
print(f"packing + upload of {nbytes / 1e9:.2f} GB: {t_up:.2f} s")
batch.close()
