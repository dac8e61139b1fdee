import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic
fam = synthetic.make_family(128, 300, dim=10, seed=20242)
coords, tensors, offsets = synthetic.pack(fam)
pairs = engine.all_pairs(128)
pc, pt = engine.pinned_empty(coords.shape, np.float64), engine.pinned_empty(tensors.shape, np.float64)
pc[...], pt[...] = coords, tensors
ctx = engine.Context(0)
for arrs, name in (((coords, tensors), "pageable"), ((pc, pt), "pinned")):
    ts = np.zeros(3)
    for it in range(22):
        ctx.synchronize()
        t0 = time.perf_counter()
        b = engine.PairBatch(ctx, arrs[0], arrs[1], offsets)
        t1 = time.perf_counter()
        b.set_pairs(pairs)
        t2 = time.perf_counter()
        ctx.synchronize()
        t3 = time.perf_counter()
        b.close()
        if it >= 2:
            ts += (t1 - t0, t2 - t1, t3 - t2)
    print(name, "create %.3f ms, set_pairs %.3f ms, sync %.3f ms" % tuple(ts / 20 * 1e3))
