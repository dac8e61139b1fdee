#!/usr/bin/env python3
"""Neighbor joining: the one-workgroup device kernel (cr_neighbor_joining_device, incl. upload of the matrix and
download of the tree) against the host implementation, per number of taxa; trees compared bit for bit.
python tools/nj_device_time.py"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from caretta_amd import engine, neighbor_joining as nj  # noqa: E402

ctx = engine.default_context()
rng = np.random.default_rng(1)
for p in (16, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 2048):
    x = rng.normal(size=(p, 6))
    d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
    d = d.max() - d + 0.0                       # shaped like max(M) - M: non-zero diagonal
    d = (d + d.T) / 2
    times = {}
    res = {}
    for dev in (False, True):
        nj.neighbor_joining(d, device=dev, ctx=ctx if dev else None)
        reps = 5 if p <= 512 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            res[dev] = nj.neighbor_joining(d, device=dev, ctx=ctx if dev else None)
        times[dev] = 1e3 * (time.perf_counter() - t0) / reps
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    print(f"P={p:5d}: host {times[False]:8.3f} ms   device {times[True]:8.3f} ms   identical={same}", flush=True)
