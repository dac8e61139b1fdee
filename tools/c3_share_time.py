"""One GPU's share of the headline configuration on 8 GPUs (every 8th of the 8 128 pairs of 128 x 300) on the library's own
layout choice, timed; the workload of tools/pmc_share.sh <tag> tools/c3_share_time.py."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic
fam = synthetic.make_family(128, 300, seed=20242)
coords, tensors, offsets = synthetic.pack(fam)
pairs = engine.all_pairs(128)[::8]            # 1016 pairs
ctx = engine.Context(0)
b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
prm = engine.make_params()
for _ in range(3): b.run(prm)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(50): b.run(prm)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 50
b.run(prm, scores_only=True)
sw, _ = b.fetch_scores()
print(f"{len(pairs)} pairs of 300 x 300, layout {b.layout()}: {dt*1e3:.3f} ms per pass, checksum {sw.sum():.6f}")
