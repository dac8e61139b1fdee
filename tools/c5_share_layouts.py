#!/usr/bin/env python3
"""One GPU's share of BASELINE config 5 on 8 GPUs (252 pairs of 1200 x 1200) on the wide layout (barrier every 8 steps, sums by
the whole workgroup) against the mid-size layout's kernels (cr_duo.h: strips paced by progress words, sums by wave 0) forced
onto it:  python tools/c5_share_layouts.py"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic
shape = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 1200, 20244, 8]     # structures, residues, seed, stride
fam = synthetic.make_family(shape[0], shape[1], seed=shape[2])
coords, tensors, offsets = synthetic.pack(fam)
pairs = engine.all_pairs(shape[0])[::shape[3]]
print(f"{len(pairs)} pairs of {shape[1]}", flush=True)
ctx = engine.Context(0)
prm = engine.make_params()
def timed(b, so):
    for _ in range(2): b.run(prm, scores_only=so)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(5): b.run(prm, scores_only=so)
    ctx.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
ref = None
for name, env in [("default (wide)", {}), ("duo 3,2,3", {"CARETTA_TEAM_PAIRS": "0", "CARETTA_STAGED": "0", "CARETTA_MID_PLAN": "3,2,3", "CARETTA_MID_PAIRS": "100000"}),
                  ("duo 3,2,4", {"CARETTA_TEAM_PAIRS": "0", "CARETTA_STAGED": "0", "CARETTA_MID_PLAN": "3,2,4", "CARETTA_MID_PAIRS": "100000"}),
                  ("duo 3,3,0", {"CARETTA_TEAM_PAIRS": "0", "CARETTA_STAGED": "0", "CARETTA_MID_PLAN": "3,3,0", "CARETTA_MID_PAIRS": "100000"}),
                  ("duo 2,2,0", {"CARETTA_TEAM_PAIRS": "0", "CARETTA_STAGED": "0", "CARETTA_MID_PLAN": "2,2,0", "CARETTA_MID_PAIRS": "100000"})]:
    for k in ("CARETTA_TEAM_PAIRS", "CARETTA_STAGED", "CARETTA_MID_PLAN", "CARETTA_MID_PAIRS"): os.environ.pop(k, None)
    os.environ.update(env)
    try:
        engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
        b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        lay = b.layout()
        full = timed(b, False); res, aln = b.fetch(); mat = timed(b, True); b.close()
    except Exception as e:
        print(name, "failed:", e); continue
    cur = (res.tobytes(), aln)
    if ref is None: ref = cur
    print(f"{name:16s} {lay}: full {full:.3f} ms, matrix only {mat:.3f} ms, identical {cur[0] == ref[0] and np.array_equal(cur[1], ref[1])}", flush=True)
