#!/usr/bin/env python3
"""What the first device operation after a long burst of kernels costs on this platform.

    python tools/stall_probe.py kernel | d2h | h2d_pageable | h2d_pinned | sleep<ms>

Runs make_pairwise_matrix for 512 x 300 (38 ms of kernels), then one small operation (two kernels of a resident
four-structure batch, or a transfer), then the same operation again.  Before every large transfer went through the
context's page-locked ring (and while the scores-only path built a 21 MB record array), the first KERNEL after the call
started on the device ~20 ms after its launch with nothing executing in between (rocprofv3 kernel + HIP trace of this
script); now it starts within 0.1 ms.  DESIGN.md section 8 has the story.
"""
import sys, time, numpy as np
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from caretta_amd import engine, multiple_alignment as ma, synthetic
mode = sys.argv[1]
fam = synthetic.make_family(512, 300, seed=20242)
msa = ma.MultipleAlignment([ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam])
prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
ctx = engine.default_context()
small = synthetic.make_family(4, 60, seed=3)
sc, st, so = synthetic.pack(small)
pc, pt = engine.pinned_empty(sc.shape, np.float64), engine.pinned_empty(st.shape, np.float64)
pc[...] = sc; pt[...] = st
resident = engine.PairBatch(ctx, sc, st, so).set_pairs(engine.all_pairs(4))
p = engine.make_params()
resident.run(p, scores_only=True); resident.fetch_scores()
msa.make_pairwise_matrix(prm)
for rep in range(3):
    msa.make_pairwise_matrix(prm)
    if mode.startswith("sleep"):
        time.sleep(float(mode[5:]) / 1000.0)
    t0 = time.perf_counter()
    if mode == "kernel" or mode.startswith("sleep"):
        resident.run(p, scores_only=True); ctx.synchronize()
    elif mode == "d2h":
        resident.fetch_scores()
    elif mode == "h2d_pageable":
        b = engine.PairBatch(ctx, sc, st, so)
    elif mode == "h2d_pinned":
        b = engine.PairBatch(ctx, pc, pt, so)
    t1 = time.perf_counter()
    # a second identical op right behind
    if mode == "kernel" or mode.startswith("sleep"):
        resident.run(p, scores_only=True); ctx.synchronize()
    elif mode == "d2h":
        resident.fetch_scores()
    elif mode.startswith("h2d"):
        b2 = engine.PairBatch(ctx, pc if mode == "h2d_pinned" else sc, pt if mode == "h2d_pinned" else st, so)
    t2 = time.perf_counter()
    print(f"{mode}: first {1e3*(t1-t0):7.3f} ms   second {1e3*(t2-t1):7.3f} ms", flush=True)
