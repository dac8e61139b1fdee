#!/usr/bin/env python3
"""Shares of a long-chain family (BASELINE config 5: 64 x 1200; one GPU's share at 2, 4, 8 GPUs = 1 008 / 504 / 252 pairs) on the
layouts that could take them: one wave per pair, the wide layout (one workgroup per pair, in rounds of 256), the row split
paced by progress words (k_pair_duo, eight waves per pair, two pairs per CU).   python tools/long_share_layouts.py [P,L,seed] [stride ...]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic
args = sys.argv[1:]
shape = [int(x) for x in args.pop(0).split(",")] if args and "," in args[0] else [64, 1200, 20244]
strides = [int(a) for a in args] or [2, 4, 8]
fam = synthetic.make_family(shape[0], shape[1], seed=shape[2])
coords, tensors, offsets = synthetic.pack(fam)
ctx = engine.Context(0)
prm = engine.make_params()
KEYS = ("CARETTA_TEAM_PAIRS", "CARETTA_STAGED", "CARETTA_MID_PLAN", "CARETTA_MID_PAIRS", "CARETTA_MID", "CARETTA_NO_TEAM", "CARETTA_TRIO")
def timed(b, so):
    for _ in range(2): b.run(prm, scores_only=so)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(5): b.run(prm, scores_only=so)
    ctx.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
for stride in strides:
    pairs = engine.all_pairs(shape[0])[::stride]
    ref = None
    for name, env in [("library's choice", {}), ("one wave per pair", {"CARETTA_NO_TEAM": "1"}),
                      ("wide, any pair count", {"CARETTA_TEAM_PAIRS": "1000000", "CARETTA_STAGED": "0", "CARETTA_MID": "0", "CARETTA_TRIO": "0"}),
                      ("duo, any pair count", {"CARETTA_TEAM_PAIRS": "0", "CARETTA_STAGED": "0", "CARETTA_MID_PAIRS": "1000000", "CARETTA_TRIO": "0"})]:
        for k in KEYS: os.environ.pop(k, None)
        os.environ.update(env)
        engine.reload_config()
        try:
            b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
            lay = b.layout()
            full = timed(b, False); res, aln = b.fetch(); mat = timed(b, True); b.close()
        except Exception as e:
            print(f"stride {stride} {name}: failed: {e}", flush=True); continue
        cur = (res.tobytes(), aln)
        if ref is None: ref = cur
        print(f"stride {stride} {len(pairs):5d} pairs of {shape[1]}  {name:22s} {lay}: full {full:.3f} ms, matrix only {mat:.3f} ms, identical {cur[0] == ref[0] and np.array_equal(cur[1], ref[1])}", flush=True)
for k in KEYS: os.environ.pop(k, None)
