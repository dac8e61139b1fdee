#!/usr/bin/env python3
"""One GPU's share of the headline configuration on 8 GPUs: every 8th of the 8 128 pairs of 128 x 300 (1 016 pairs).

    python tools/c3_share.py [--family=P,L,seed] [stride ...]     default 128,300,20242 and stride 8; "2 4 8" = the shares at 2, 4, 8 GPUs
    C3_LIMIT=1 / C3_ONE=1 / C3_ALL=1 python tools/c3_share.py ...   other mode sets (see MODES)

For every share: one wave per pair, the library's own choice, and strip plans of the mid-size layout (cr_duo.h) forced through
the environment (CARETTA_MID_PLAN=RA,RB,nA is read by cr_batch_set_pairs); full pipeline and matrix entries only, all results
compared with the single-wave path bit for bit.
"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from caretta_amd import engine, synthetic  # noqa: E402

KEYS = ("CARETTA_TRIO", "CARETTA_TRIO_PAIRS", "CARETTA_TRIO_FROM", "CARETTA_TRIO_WAVES", "CARETTA_TRIO_WAVES2", "CARETTA_MID_ANY", "CARETTA_MID_PLAN", "CARETTA_MID_LDS_KB", "CARETTA_MID_PAIRS", "CARETTA_MID", "CARETTA_WIDE", "CARETTA_TEAM_PAIRS",
        "CARETTA_STAGED", "CARETTA_STAGED_WAVES", "CARETTA_NO_TEAM", "CARETTA_NO_WIDE")
FORCE = {"CARETTA_MID_PAIRS": "100000", "CARETTA_TRIO": "0"}
MODES = [("single wave", {"CARETTA_NO_TEAM": "1", "CARETTA_MID": "0", "CARETTA_TRIO": "0"}),
         ("default", {}),
         ("trio 1 + 2 waves", {"CARETTA_TRIO_PAIRS": "100000", "CARETTA_TRIO_WAVES": "3"}),
         ("trio 1 + 3 waves", {"CARETTA_TRIO_PAIRS": "100000", "CARETTA_TRIO_WAVES": "4"}),
         ("mid 3,2,1 (2 waves)", dict(FORCE, CARETTA_MID_PLAN="3,2,1")),
         ("mid 3,3,0 (2 waves)", dict(FORCE, CARETTA_MID_PLAN="3,3,0")),
         ("mid 2,1,2 (3 waves)", dict(FORCE, CARETTA_MID_PLAN="2,1,2")),
         ("mid 2,2,0 (3 waves)", dict(FORCE, CARETTA_MID_PLAN="2,2,0")),
         ("mid 2,1,1 (4 waves)", dict(FORCE, CARETTA_MID_PLAN="2,1,1")),
         ("mid 1,1,0 (5 waves)", dict(FORCE, CARETTA_MID_PLAN="1,1,0")),
         ("mid 3,2,1, 70 KB (2 pairs per CU)", dict(FORCE, CARETTA_MID_PLAN="3,2,1", CARETTA_MID_LDS_KB="70"))]
if os.environ.get("C3_TRIO"):           # only the split by function, one to four score waves, against one wave per pair and the library's choice
    MODES = MODES[:2] + [(f"trio 1 + {w - 1} waves", {"CARETTA_TRIO_PAIRS": "100000", "CARETTA_TRIO_WAVES": str(w), "CARETTA_STAGED": "0"}) for w in (2, 3, 4, 5)]
if os.environ.get("C3_STAGES"):         # score waves per STAGE (seed + its waves, alignment + its waves), against one wave per pair and the library's choice
    MODES = MODES[:2] + [(f"trio 1+{a - 1} / 1+{b - 1}", {"CARETTA_TRIO_PAIRS": "100000", "CARETTA_TRIO_WAVES": str(a), "CARETTA_TRIO_WAVES2": str(b), "CARETTA_STAGED": "0"})
                         for a, b in ((2, 2), (3, 2), (3, 3), (4, 2), (4, 3), (4, 4), (5, 3), (5, 4))]
if os.environ.get("C3_FEW"):            # at most 256 pairs: the one-pair-per-CU layouts and staged scores against the split by function forced onto the list
    MODES = MODES[:2] + [("no trio", {"CARETTA_TRIO": "0"})] + [(f"trio 1 + {w - 1} waves", {"CARETTA_TRIO_PAIRS": "100000", "CARETTA_TEAM_PAIRS": "0", "CARETTA_TRIO_FROM": "0", "CARETTA_TRIO_WAVES": str(w), "CARETTA_STAGED": "0"}) for w in (3, 4, 5)]
if os.environ.get("C3_LIMIT"):          # where the layout stops paying: the library's plan against one wave per pair
    MODES = MODES[:1] + [("mid 3,2,1", dict(FORCE, CARETTA_MID_PLAN="3,2,1")), ("trio 1 + 2 waves", {"CARETTA_TRIO_PAIRS": "100000", "CARETTA_TRIO_WAVES": "3"}),
                         ("trio 1 + 3 waves", {"CARETTA_TRIO_PAIRS": "100000", "CARETTA_TRIO_WAVES": "4"})]
if os.environ.get("C3_ONE"):            # k_pair_duo with ONE strip against k_seed + k_align (the kernel itself, no pacing)
    MODES = MODES[:2] + [("mid 3,3,0 one wave", dict(FORCE, CARETTA_MID_PLAN="3,3,0", CARETTA_MID_ANY="1", CARETTA_TEAM_PAIRS="0", CARETTA_STAGED="0"))]
if os.environ.get("C3_ALL"):            # the one-pair-per-CU layouts forced onto the list
    MODES += [("wide 3,3 (2 waves)", {"CARETTA_WIDE": "3,3,0,8"}),
              ("wide 2,2 (3 waves)", {"CARETTA_WIDE": "2,2,0,8"}),
              ("staged (5 waves)", {"CARETTA_STAGED_WAVES": str(1 << 40)})]


def timed(batch, ctx, prm, reps, scores_only):
    for _ in range(3):
        batch.run(prm, scores_only=scores_only)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.run(prm, scores_only=scores_only)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    args = sys.argv[1:]
    num, length, seed = 128, 300, 20242
    if args and args[0].startswith("--family="):                  # e.g. --family=32,150,20241
        num, length, seed = (int(x) for x in args.pop(0).split("=")[1].split(","))
    strides = [int(a) for a in args] or [8]
    fam = synthetic.make_family(num, length, seed=seed)
    coords, tensors, offsets = synthetic.pack(fam)
    ctx = engine.Context(0)
    prm = engine.make_params()
    for stride in strides:
        pairs = engine.all_pairs(num)[::stride]
        ref = None
        for name, env in MODES:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                engine.reload_config()        # (the library reads its calibration switches once: cr_config.h)
                b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
                lay = b.layout()
                full = timed(b, ctx, prm, 20, False)
                res, aln = b.fetch()
                mat = timed(b, ctx, prm, 20, True)
                sc = b.fetch_scores()
                b.close()
            except Exception as e:  # noqa: BLE001
                print(f"stride {stride} {name:22s}: {type(e).__name__}: {e}", flush=True)
                continue
            cur = (res.tobytes(), aln, sc[0])
            if name == "single wave":
                ref = cur
            same = ""
            if ref is not None:
                same = "identical" if (cur[0] == ref[0] and np.array_equal(cur[1], ref[1]) and np.array_equal(cur[2], ref[2])) else "DIFFERENT"
            print(f"stride {stride} {len(pairs):5d} pairs  {name:22s}: full {full:7.3f} ms   matrix only {mat:7.3f} ms   {same}  {lay[0]} {lay[1]}", flush=True)
    for k in KEYS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
