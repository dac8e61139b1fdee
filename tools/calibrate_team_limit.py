#!/usr/bin/env python3
"""Single-wave vs team kernels (4 waves per pair) as a function of the number of pairs: python tools/calibrate_team_limit.py LIMIT"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["CARETTA_TEAM_PAIRS"] = sys.argv[1]
from caretta_amd import engine, synthetic  # noqa: E402

ctx = engine.Context(0)
out = []
for L in (150, 300):
    for num in (12, 17, 23, 32, 40, 46, 56, 64):
        fam = synthetic.make_family(num, L, seed=L + num, clades=4)
        coords, tensors, offsets = synthetic.pack(fam)
        engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
        b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(engine.all_pairs(num))
        prm = engine.make_params()
        for _ in range(3):
            b.run(prm)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            b.run(prm)
        ctx.synchronize()
        out.append(f"L{L}/{num * (num - 1) // 2}:{(time.perf_counter() - t0) / 10 * 1e3:.3f}")
        b.close()
print("team limit " + sys.argv[1] + ": " + " ".join(out))
