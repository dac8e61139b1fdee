#!/usr/bin/env python3
"""Ragged pair lists -- what every real input is (the reference's example: 85 / 79 / 80 residues; its pair loop,
multiple_alignment.py:158-170, makes no length assumption):

  ragged : 160 structures of 80 .. 520 residues (one family, seeded), all 12 720 pairs: ms per pass and Mcells/s of the full
           pipeline and of the matrix entries alone, beside the equal-length headline's Mcells/s;
  mixed  : 30 structures of 150 residues + 2 of 600 (496 pairs) as ONE list -- split into size classes by cr_batch_set_pairs,
           and with CARETTA_CLASSES=0 as the one list it used to be -- against the sum of its homogeneous parts run alone
           (150 x 150, 150 x 600 / 600 x 150, 600 x 600).

    python tools/ragged_time.py            (prints one JSON record per experiment; bench.py imports `ragged_record`, `mixed_record`)
"""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from caretta_amd import engine, synthetic  # noqa: E402


def timed(batch, ctx, prm, reps, scores_only=False, warm=3):
    for _ in range(warm):
        batch.run(prm, scores_only=scores_only)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run(prm, scores_only=scores_only)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


def ragged_record(ctx, prm, oracle=None, reps=10, check_frac=0.01):
    fam = synthetic.make_ragged_family(160, 80, 520, seed=20250)
    coords, tensors, offsets = synthetic.pack(fam)
    lengths = np.diff(offsets)
    pairs = engine.all_pairs(len(fam))
    cells = float((lengths[pairs[:, 0]] * lengths[pairs[:, 1]]).sum())
    b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
    t_full = timed(b, ctx, prm, reps)
    res, aln = b.fetch()
    t_mat = timed(b, ctx, prm, reps, scores_only=True)
    sw, _ = b.fetch_scores()
    rec = {"structures": len(fam), "residues": [int(lengths.min()), int(lengths.max())], "pairs": int(len(pairs)), "cells": cells,
           "ms": t_full * 1e3, "pairs_per_s": len(pairs) / t_full, "mcells_per_s": cells / t_full / 1e6,
           "matrix_only_ms": t_mat * 1e3, "matrix_only_mcells_per_s": cells / t_mat / 1e6,
           "layout": b.layout()[0], "classes": [list(x) for x in b.part_layouts()],
           "scores_equal_full_run": bool(np.array_equal(sw, res["sw"]))}
    if oracle is not None:
        rng = np.random.default_rng(5)
        pick = np.sort(rng.choice(len(pairs), size=max(int(np.ceil(check_frac * len(pairs))), 128), replace=False))
        ref, ref_aln = oracle.pairwise_batch(coords, tensors, offsets, pairs[pick], want_aln=True, nthreads=max(1, min(oracle.max_threads(), os.cpu_count() or 1)))
        bad = 0
        for k, p in enumerate(pick):
            ln = int(ref["aln_len"][k])
            ok = int(res["aln_len"][p]) == ln and np.array_equal(aln[p, :, :ln], ref_aln[k, :, :ln])
            ok = ok and all(np.array_equal(res[key][p], ref[key][k]) for key in ("sw", "dtw_score", "rmsd", "tm", "coverage"))
            bad += 0 if ok else 1
        rec["pair_gate"] = {"checked": int(len(pick)), "fraction": len(pick) / len(pairs), "mismatches": bad,
                            "what": "alignment rows and lengths exact; sw, dtw_score, rmsd, tm, coverage bit-identical to the C oracle"}
    b.close()
    return rec


def mixed_record(ctx, prm, reps=20):
    fam = synthetic.make_mixed_family(30, 150, 2, 600, seed=20251)
    coords, tensors, offsets = synthetic.pack(fam)
    lengths = np.diff(offsets)
    pairs = engine.all_pairs(len(fam))
    out = {"structures": "30 x 150 + 2 x 600", "pairs": int(len(pairs))}
    results = {}
    for name, env in (("classes", {}), ("one_list", {"CARETTA_CLASSES": "0"})):
        os.environ.pop("CARETTA_CLASSES", None)
        os.environ.update(env)
        engine.reload_config()
        b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        t_full = timed(b, ctx, prm, reps)
        res, aln = b.fetch()
        t_mat = timed(b, ctx, prm, reps, scores_only=True)
        out[name] = {"ms": t_full * 1e3, "matrix_only_ms": t_mat * 1e3, "layout": b.layout()[0], "classes": [list(x) for x in b.part_layouts()]}
        results[name] = (res.tobytes(), aln)
        b.close()
    os.environ.pop("CARETTA_CLASSES", None)
    engine.reload_config()
    out["bit_identical"] = bool(results["classes"][0] == results["one_list"][0] and np.array_equal(results["classes"][1], results["one_list"][1]))
    # the homogeneous parts, each as a list of its own, one after the other
    n, m = lengths[pairs[:, 0]], lengths[pairs[:, 1]]
    parts = {"150x150": (n == 150) & (m == 150), "150x600": (n != m), "600x600": (n == 600) & (m == 600)}
    total = total_mat = 0.0
    out["parts"] = {}
    for name, mask in parts.items():
        sub = np.ascontiguousarray(pairs[mask])
        if not len(sub):
            continue
        b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(sub)
        t_full = timed(b, ctx, prm, reps)
        t_mat = timed(b, ctx, prm, reps, scores_only=True)
        out["parts"][name] = {"pairs": int(len(sub)), "ms": t_full * 1e3, "matrix_only_ms": t_mat * 1e3, "layout": b.layout()[0]}
        total += t_full
        total_mat += t_mat
        b.close()
    out["sum_of_parts_ms"] = total * 1e3
    out["sum_of_parts_matrix_only_ms"] = total_mat * 1e3
    out["classes_over_sum_of_parts"] = out["classes"]["ms"] / (total * 1e3)
    out["one_list_over_sum_of_parts"] = out["one_list"]["ms"] / (total * 1e3)
    return out


if __name__ == "__main__":
    ctx = engine.Context(0)
    prm = engine.make_params()
    orc = None
    if "--gate" in sys.argv:
        from oracle.pyoracle import Oracle
        orc = Oracle()
    print(json.dumps({"ragged": ragged_record(ctx, prm, orc)}), flush=True)
    print(json.dumps({"mixed": mixed_record(ctx, prm)}), flush=True)
