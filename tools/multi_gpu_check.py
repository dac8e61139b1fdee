#!/usr/bin/env python3
"""The single-process multi-GPU path (cr_multi_*: one context per visible GPU, one grouped RCCL all-gather) against the
one-GPU path on BASELINE config 4 (512 x 300) or the given P x L: same scores bit for bit, and the time of both.

    python tools/multi_gpu_check.py [P [L]]          prints ONE JSON line

bench.py runs this as a CHILD process (with a time limit) when more than one GPU is visible to a one-rank run, so that a
fault on hardware this path has never seen cannot take the benchmark line with it.
"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from caretta_amd import engine, synthetic  # noqa: E402


def main():
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    fam = synthetic.make_family(num, length, dim=10, seed=20243)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(num)
    prm = engine.make_params()
    ctx = engine.Context(0)
    reps = 3
    for it in range(reps + 1):                         # (both sides: upload, pair list, kernels, download, every call)
        if it == 1:
            t0 = time.perf_counter()
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(prm, scores_only=True)
        sw_one, flags_one = batch.fetch_scores()
        batch.close()
    t_one = (time.perf_counter() - t0) / reps
    multi = engine.MultiDevice()
    for it in range(reps + 1):
        if it == 1:
            t0 = time.perf_counter()
        sw, flags = multi.pairwise_scores(coords, tensors, offsets, prm)
    t_multi = (time.perf_counter() - t0) / reps
    # the consumer of the matrix: identical guide trees (neighbor joining is 1-ulp sensitive, neighbor_joining.py:118-129)
    from caretta_amd import neighbor_joining as nj
    m_one, m_multi = engine.assemble_matrix(pairs, sw_one, num), engine.assemble_matrix(pairs, sw, num)
    tree_one, bl_one = nj.neighbor_joining(m_one.max() - m_one)
    tree_multi, bl_multi = nj.neighbor_joining(m_multi.max() - m_multi)
    trees_identical = bool(np.array_equal(tree_one, tree_multi) and np.array_equal(bl_one, bl_multi))
    out = {"devices": multi.num_devices, "structures": num, "residues": length, "pairs": int(len(pairs)),
           "one_gpu_ms": t_one * 1e3, "multi_gpu_ms": t_multi * 1e3, "speedup": t_one / t_multi,
           "last_call_ms": dict(zip(("slowest_share_events", "all_gather_events", "download_events_plus_host_scatter"), multi.last_ms())),
           "scores_identical": bool(np.array_equal(sw, sw_one) and np.array_equal(flags, flags_one)),
           "nj_trees_identical": trees_identical,
           "note": "one process, one context + host thread per GPU, cr_partition_pairs, one grouped ncclAllGather (RCCL bound at run "
                   "time); both times include the upload of the structures and the download of the score vector"}
    multi.close()
    print(json.dumps(out))
    if not (out["scores_identical"] and trees_identical):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
