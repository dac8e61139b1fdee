#!/usr/bin/env python3
"""Time one pass over 4096 equal-length pairs for every rows-per-lane setting (CARETTA_FORCE_R) and length: the data
behind rows_per_lane() in cr_api.hip.  Run each R in its own process: python tools/calibrate_rows_per_lane.py R"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
R = sys.argv[1]
os.environ["CARETTA_FORCE_R"] = R
from caretta_amd import engine, synthetic  # noqa: E402

ctx = engine.Context(0)
out = []
for L in (48, 64, 80, 100, 128, 150, 192, 220, 256, 300, 320, 350, 384, 420, 450, 512, 560, 600, 640, 700, 800, 960):
    fam = synthetic.make_family(91, L, seed=L, clades=4)          # 4095 pairs
    coords, tensors, offsets = synthetic.pack(fam)
    engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
    b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(engine.all_pairs(len(fam)))
    prm = engine.make_params()
    for _ in range(2):
        b.run(prm)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b.run(prm)
    ctx.synchronize()
    out.append((L, (time.perf_counter() - t0) / 5 * 1e3))
    b.close()
print("R=" + R + " " + " ".join(f"{L}:{ms:.3f}" for L, ms in out))
