import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic
rng = np.random.default_rng(3)
fam = []
for k in range(160):
    L = int(rng.integers(80, 520))
    fam.append(synthetic.make_family(1, L, seed=1000 + k, clades=1)[0])
coords, tensors, offsets = synthetic.pack(fam)
ctx = engine.Context(0)
pairs = engine.all_pairs(len(fam))
b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
prm = engine.make_params()
for _ in range(3): b.run(prm)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(10): b.run(prm)
ctx.synchronize()
dt = (time.perf_counter() - t0) / 10
sw, fl = b.fetch_scores()
print(f"{len(pairs)} ragged pairs (80..520 residues): {dt*1e3:.2f} ms per pass, checksum {sw.sum():.6f}")
