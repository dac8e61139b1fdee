import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic
fam = synthetic.make_family(64, 1200, seed=20244)
coords, tensors, offsets = synthetic.pack(fam)
pairs = engine.all_pairs(64)[::8]            # 252 pairs: one GPU's share of config 5 on 8 GPUs
ctx = engine.Context(0)
b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
prm = engine.make_params()
for _ in range(2): b.run(prm)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5): b.run(prm)
ctx.synchronize()
sw, _ = b.fetch_scores()
print(f"{len(pairs)} pairs of 1200 x 1200: {(time.perf_counter()-t0)/5*1e3:.2f} ms per pass, checksum {sw.sum():.6f}")
for _ in range(2): b.run(prm, scores_only=True)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5): b.run(prm, scores_only=True)
ctx.synchronize()
sw2, _ = b.fetch_scores()
print(f"{len(pairs)} pairs of 1200 x 1200, matrix entries only (cr_batch_run_scores): {(time.perf_counter()-t0)/5*1e3:.2f} ms per pass, "
      f"scores {'identical' if np.array_equal(sw, sw2) else 'DIFFER'}")
