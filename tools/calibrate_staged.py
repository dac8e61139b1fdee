#!/usr/bin/env python3
"""Pair batches on staged scores (cr_staged.h) against the fused kernels.

    python tools/calibrate_staged.py [workload ...]      workloads: those of tools/calibrate_wide.py

For every workload: time the batch with CARETTA_STAGED=0 (the fused kernels the library would otherwise choose) and
with the staged path forced (CARETTA_STAGED_WAVES very large), full pipeline and matrix entries alone, and demand
bit-identical results (all PairResult fields and the alignment rows).
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import numpy as np  # noqa: E402

from caretta_amd import engine, synthetic  # noqa: E402
from calibrate_wide import WORKLOADS  # noqa: E402


def timed(batch, ctx, prm, reps, scores_only):
    for _ in range(2):
        batch.run(prm, scores_only=scores_only)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run(prm, scores_only=scores_only)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    names = sys.argv[1:] or ["c2", "c2half", "one300", "p64x300", "p120x450"]
    ctx = engine.Context(0)
    prm = engine.make_params()
    for name in names:
        num, length, seed, stride = WORKLOADS[name]
        fam = synthetic.make_family(num, length, seed=seed)
        coords, tensors, offsets = synthetic.pack(fam)
        pairs = engine.all_pairs(num)[::stride]
        out = {}
        os.environ["CARETTA_TRIO"] = "0"             # (the split by function of cr_trio.h would take the lists of 65 .. 320 rows in both modes)
        for mode in ("fused", "staged"):
            os.environ.pop("CARETTA_STAGED", None)
            os.environ.pop("CARETTA_STAGED_WAVES", None)
            if mode == "fused":
                os.environ["CARETTA_STAGED"] = "0"
            else:
                os.environ["CARETTA_STAGED_WAVES"] = str(1 << 40)
            engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
            b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
            reps = 20 if len(pairs) < 3000 else 5
            full = timed(b, ctx, prm, reps, False)
            res, aln = b.fetch()
            sc = timed(b, ctx, prm, reps, True)
            scores = b.fetch_scores()
            out[mode] = (full, sc, res, aln, scores)
            b.close()
        same = (out["fused"][2].tobytes() == out["staged"][2].tobytes() and np.array_equal(out["fused"][3], out["staged"][3])
                and np.array_equal(out["fused"][4], out["staged"][4]))
        print(f"{name:10s} {len(pairs):6d} pairs of {length:5d}: fused {out['fused'][0]:8.3f} ms (matrix only {out['fused'][1]:7.3f})   "
              f"staged {out['staged'][0]:8.3f} ms (matrix only {out['staged'][1]:7.3f})   identical: {same}", flush=True)


if __name__ == "__main__":
    main()
