import sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import multiple_alignment as ma, neighbor_joining as nj, synthetic, _capi
from caretta_amd.engine import default_context, make_params
from caretta_amd._capi import ptr, check
num, length = int(sys.argv[1]), int(sys.argv[2])
fam = synthetic.make_family(num, length, seed=20242)
prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
msa = ma.MultipleAlignment(prots)
prm = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
m = msa.make_pairwise_matrix(prm)
tree, _ = nj.neighbor_joining(m.max() - m)
lib = _capi.load()
coords, tensors, offsets = ma.pack_proteins(prots)
p = make_params(gamma_tensor=7.0, gamma_coords=0.03, gap_open=1.0, gap_extend=0.01)
tree_u = np.ascontiguousarray(tree, dtype=np.uint64)
for rep in range(3):
    t0 = time.perf_counter()
    h = C.c_void_p()
    check(lib.cr_progressive_align(default_context()._h, ptr(coords), ptr(tensors), ptr(offsets), num, 10, ptr(tree_u), tree_u.shape[0], C.byref(p), 1.0, 1.0, C.byref(h)))
    t1 = time.perf_counter()
    sizes = np.zeros(5, np.int64); check(lib.cr_progressive_sizes(h, ptr(sizes)))
    msa_m = np.zeros((num, int(sizes[0])), np.int64); check(lib.cr_progressive_fetch_msa(h, ptr(msa_m)))
    t2 = time.perf_counter()
    total = int(sizes[2])
    aln = np.zeros(2 * total, np.int64); xn, tn, wn = np.zeros((total, 3)), np.zeros((total, 10)), np.zeros(total)
    check(lib.cr_progressive_fetch_nodes(h, ptr(aln), ptr(xn), ptr(tn), ptr(wn)))
    t3 = time.perf_counter()
    lib.cr_progressive_destroy(h)
    t4 = time.perf_counter()
    msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, prm, dict(flexible=False, verbose=False))
    t5 = time.perf_counter()
    print(f"P={num}: C align {1e3*(t1-t0):.2f} ms, msa {1e3*(t2-t1):.2f}, fetch nodes {1e3*(t3-t2):.2f}, destroy {1e3*(t4-t3):.2f}; python progressive_align total {1e3*(t5-t4):.2f}; levels {sizes[3]}")
