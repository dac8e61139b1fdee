import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import multiple_alignment as ma, neighbor_joining as nj, synthetic
for num, L in ((64, 120), (64, 180), (128, 150)):
    fam = synthetic.make_family(num, L, seed=20242)
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
    msa = ma.MultipleAlignment(prots)
    prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
    m = msa.make_pairwise_matrix(prm)
    tree, _ = nj.neighbor_joining(m.max() - m)
    msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
    t0 = time.perf_counter()
    for _ in range(3): msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
    print(num, L, f"{(time.perf_counter()-t0)/3*1e3:.2f} ms, levels {int(msa.node_table[:,3].max())}")
