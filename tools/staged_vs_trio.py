"""Short pair lists of one-strip structures: staged scores (cr_staged.h) against the split by function (cr_trio.h), full
pipeline and matrix entries only, around the pair counts from which the layout table prefers the split by function
(65 / 111 / 161 pairs for up to 192 / 256 / 320 rows).   python tools/staged_vs_trio.py"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic


def timed(batch, ctx, prm, scores_only, reps=30):
    for _ in range(3):
        batch.run(prm, scores_only=scores_only)
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        batch.run(prm, scores_only=scores_only)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


ctx = engine.Context(0)
prm = engine.make_params()
for rows in (100, 150, 200, 250, 300):
    fam = synthetic.make_family(40, rows, seed=4100 + rows)
    coords, tensors, offsets = synthetic.pack(fam)
    allp = engine.all_pairs(40)
    for npairs in (48, 64, 80, 96, 128, 160, 192, 256, 320):
        pairs = allp[:: max(1, len(allp) // npairs)][:npairs]
        out = []
        for mode, env in (("staged", {"CARETTA_TRIO": "0", "CARETTA_STAGED_WAVES": str(1 << 40)}), ("by function", {"CARETTA_TRIO_FROM": "1", "CARETTA_STAGED": "0"}),
                          ("library", {})):
            for k in ("CARETTA_TRIO", "CARETTA_STAGED_WAVES", "CARETTA_TRIO_FROM", "CARETTA_STAGED"):
                os.environ.pop(k, None)
            os.environ.update(env)
            engine.reload_config()
            b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
            out.append((mode, b.layout()[0], timed(b, ctx, prm, False), timed(b, ctx, prm, True)))
            b.close()
        print(f"{rows:4d} rows {npairs:4d} pairs: " + "   ".join(f"{m} [{lay}] {full:.3f} / {sc:.3f} ms" for m, lay, full, sc in out))
