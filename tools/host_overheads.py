#!/usr/bin/env python3
"""Where the wall time of MultipleAlignment.make_pairwise_matrix goes on the host side of the C ABI.

    python tools/host_overheads.py [P] [L]
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from caretta_amd import multiple_alignment as ma, synthetic  # noqa: E402
from caretta_amd.engine import PairBatch, all_pairs, assemble_matrix, default_context, make_params  # noqa: E402


def main():
    num = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    fam = synthetic.make_family(num, length, seed=20242)
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
    msa = ma.MultipleAlignment(prots)
    ctx = default_context()
    params = make_params(gamma_tensor=7.0, gamma_coords=0.03)
    for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
        t = [time.perf_counter()]
        coords, tensors, offsets = ma.pack_proteins(msa.sequences)
        t.append(time.perf_counter())
        batch = PairBatch(ctx, coords, tensors, offsets)
        t.append(time.perf_counter())
        pairs = all_pairs(num)
        batch.set_pairs(pairs)
        t.append(time.perf_counter())
        batch.run(params, scores_only=True)        # what make_pairwise_matrix runs
        ctx.synchronize()
        t.append(time.perf_counter())
        sw, flags = batch.fetch_scores()
        t.append(time.perf_counter())
        m = assemble_matrix(pairs, sw, num)
        t.append(time.perf_counter())
        batch.close()
        t.append(time.perf_counter())
        names = ["pack", "create+upload", "set_pairs (scratch alloc)", "run+sync", "fetch results", "assemble", "destroy"]
        print(f"P={num} L={length} rep {rep}: " + ", ".join(f"{n} {1e3 * (b - a):.2f}" for n, a, b in zip(names, t, t[1:]))
              + f" | total {1e3 * (t[-1] - t[0]):.2f} ms, checksum {m.sum():.6f}")


if __name__ == "__main__":
    main()
