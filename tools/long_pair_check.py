import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic
ctx = engine.Context(0)
for L in (5000, 15000, 30000):
    fam = synthetic.make_family(2, L, seed=5, clades=1)
    coords, tensors, offsets = synthetic.pack(fam)
    try:
        t0 = time.time()
        b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(np.array([[0, 1]], np.int32))
        b.run(engine.make_params())
        res, aln = b.fetch()
        dt = time.time() - t0
        ln = int(res["aln_len"][0])
        a = aln[0, :, :ln]
        ok = all(np.array_equal(r[r >= 0], np.arange(L)) for r in a)
        print(L, f"{dt:.2f} s", "len", ln, "rows complete", ok, "flags", res["flags"][0], "sw", res["sw"][0], "rmsd", res["rmsd"][0])
        b.close()
    except Exception as e:
        print(L, type(e).__name__, str(e)[:200])
