#!/usr/bin/env python3
"""End-to-end timing of the path and its immediate consumers on one MI355X:
make_pairwise_matrix (batched GPU) -> max - M -> neighbor_joining (host C++ below 256 structures, the one-launch device kernel from there) -> progressive_align (GPU, whole tree resident, one launch pair per tree level).

    python tools/bench_msa.py [P] [L]
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from caretta_amd import multiple_alignment as ma, neighbor_joining as nj, synthetic  # noqa: E402


def main():
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    fam = synthetic.make_family(num, length, seed=20242)
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    msa.make_pairwise_matrix(prm)                      # warm-up (context, library load)
    t0 = time.perf_counter()
    m = msa.make_pairwise_matrix(prm)
    t1 = time.perf_counter()
    d = m.max() - m
    tree, _ = nj.neighbor_joining(d)
    t2 = time.perf_counter()
    aln = msa.multiple_align(d, gap_open_penalty=1.0, gap_extend_penalty=0.01, consensus_weight=1.0, gamma_weight=1.0,
                             score_function_params=prm, mean_function_params=dict(flexible=False, verbose=False))
    t3e = time.perf_counter()
    alone = []
    for _ in range(7):                                 # (every call first returns the previous call's nodes to the library's cache)
        t3 = time.perf_counter()
        msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
        alone.append(1e3 * (time.perf_counter() - t3))
    levels = int(msa.node_table[:, 3].max())
    width = len(next(iter(aln.values())))
    print(f"P={num} L={length}: pairwise matrix {1e3 * (t1 - t0):.1f} ms (incl. upload/download), "
          f"neighbor joining {1e3 * (t2 - t1):.1f} ms, NJ+progressive alignment {1e3 * (t3e - t2):.1f} ms "
          f"({num - 1} nodes), progressive alignment alone {np.median(alone):.1f} ms (median of 7 calls: "
          f"{min(alone):.2f} ... {max(alone):.2f}) in {levels} tree levels, MSA width {width}")


if __name__ == "__main__":
    main()
