#!/usr/bin/env python3
"""Wide kernels (one wave per strip, up to 16 waves per pair) against the default kernel choice.

    python tools/calibrate_wide.py [workload ...]      workloads: c5share c2 c5 one300 p64x300 p120x600

For every workload: time the default path (the library's own choice) and the path WITHOUT the wide kernels
(CARETTA_NO_WIDE: four-wave teams or one wave per pair), then strip plans of the wide kernels -- RA rows per lane in the
first nA strips, RB in the others, a barrier every B steps (CARETTA_WIDE=RA,RB,nA,B is read by cr_batch_set_pairs) -- and
demand bit-identical results (all PairResult fields and the alignment rows) against the path without wide kernels, which
tests/test_gpu_parity.py pins to the oracle.
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from caretta_amd import engine, synthetic  # noqa: E402

WORKLOADS = {
    "c5share": (64, 1200, 20244, 8),      # 252 pairs: one GPU's share of BASELINE config 5 on 8 GPUs
    "c5": (64, 1200, 20244, 1),
    "c3share": (128, 300, 20242, 8),      # 1016 pairs: one GPU's share of the headline configuration on 8 GPUs
    "c3quarter": (128, 300, 20242, 4),    # 2032 pairs
    "c2": (32, 150, 20241, 1),
    "c2half": (32, 150, 20241, 2),        # 248 pairs: at most one wave per CU
    "c2x4": (64, 150, 20241, 1),          # 2016 pairs: two per SIMD
    "one300": (2, 300, 7, 1),
    "p64x300": (12, 300, 11, 1),          # 66 pairs
    "p120x600": (16, 600, 12, 1),         # 120 pairs
    "c4share": (512, 300, 20243, 8),      # 16352 pairs: one GPU's share of config 4
    "p120x900": (16, 900, 13, 1),
    "p105x1500": (15, 1500, 14, 1),
    "p120x450": (16, 450, 15, 1),
    "p120x750": (16, 750, 16, 1),
    "p28x750": (8, 750, 17, 1),
    "p28x1000": (8, 1000, 18, 1),
    "p28x1500": (8, 1500, 19, 1),
    "p6x2000": (4, 2000, 20, 1),
}


def grid(length):
    plans = [(2, 2, 0, 8), (3, 3, 0, 8), (2, 2, 0, 2), (3, 3, 0, 2)]
    for na in range(1, 8):
        if na * 192 < length:
            plans.append((3, 2, na, 8))
    return plans


def timed_scores(batch, ctx, prm, reps):
    for _ in range(2):
        batch.run(prm, scores_only=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run(prm, scores_only=True)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def timed(batch, ctx, prm, reps):
    for _ in range(2):
        batch.run(prm)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run(prm)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def same(a, b):
    ra, aa = a
    rb, ab = b
    for name in ra.dtype.names:
        if not np.array_equal(ra[name], rb[name]):
            return f"field {name} differs"
    for p in range(len(ra)):
        ln = int(ra["aln_len"][p])
        if not np.array_equal(aa[p, :, :ln], ab[p, :, :ln]):
            return f"alignment of pair {p} differs"
    return None


def main():
    names = sys.argv[1:] or ["c5share", "c2"]
    ctx = engine.Context(0)
    prm = engine.make_params()
    for name in names:
        num, length, seed, stride = WORKLOADS[name]
        fam = synthetic.make_family(num, length, seed=seed)
        coords, tensors, offsets = synthetic.pack(fam)
        pairs = engine.all_pairs(num)[::stride]
        reps = 5 if length >= 600 else 20
        os.environ.pop("CARETTA_WIDE", None)
        engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
        b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        print(f"{name}: {len(pairs)} pairs of {length}: default {timed(b, ctx, prm, reps):.3f} ms, scores only {timed_scores(b, ctx, prm, reps):.3f} ms", flush=True)
        dflt = b.fetch() if False else None
        b.run(prm)
        dflt = b.fetch()
        os.environ["CARETTA_NO_WIDE"] = "1"
        engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
        b.set_pairs(pairs)
        base_ms = timed(b, ctx, prm, reps)
        base = b.fetch()
        os.environ.pop("CARETTA_NO_WIDE", None)
        verdict = same(base, dflt)
        print(f"  without wide kernels: {base_ms:.3f} ms; default path {'bit-identical' if verdict is None else 'MISMATCH: ' + verdict}", flush=True)
        for ra, rb, na, sync in grid(length):
            os.environ["CARETTA_WIDE"] = f"{ra},{rb},{na},{sync}"
            engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
            b.set_pairs(pairs)
            ms = timed(b, ctx, prm, reps)
            ctx.set_profiling(reps)
            for _ in range(reps):
                b.run(prm)
            st, _ = b.stage_ms()
            ctx.set_profiling(0)
            verdict = same(base, b.fetch())
            ms_sc = timed_scores(b, ctx, prm, reps)
            sw_sc, _ = b.fetch_scores()
            if verdict is None and not np.array_equal(sw_sc, base[0]["sw"]):
                verdict = "scores-only run differs"
            print(f"  RA={ra} RB={rb} nA={na} B={sync:2d}: {ms:.3f} ms (k_seed {st[0]:.3f} k_align {st[1]:.3f}), scores only {ms_sc:.3f} ms  "
                  f"{'bit-identical' if verdict is None else 'MISMATCH: ' + verdict}", flush=True)
        os.environ.pop("CARETTA_WIDE", None)
        b.close()


if __name__ == "__main__":
    main()
