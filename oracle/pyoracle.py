"""ctypes front-end of the C oracle -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under caretta_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


class Params(C.Structure):
    _fields_ = [("gamma_tensor", C.c_double), ("gamma_coords", C.c_double), ("gap_open", C.c_double),
                ("gap_extend", C.c_double), ("sw_gap", C.c_double)]


class PairOut(C.Structure):
    _fields_ = [("sw", C.c_double), ("dtw_score", C.c_double), ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("rmsd", C.c_double), ("coverage", C.c_double), ("tm", C.c_double), ("seed_score", C.c_double),
                ("aln_len", C.c_int64), ("seed_len", C.c_int64), ("flags", C.c_uint32)]


PAIR_DTYPE = np.dtype([("sw", "f8"), ("dtw_score", "f8"), ("R", "f8", (9,)), ("t", "f8", (3,)), ("rmsd", "f8"),
                       ("coverage", "f8"), ("tm", "f8"), ("seed_score", "f8"), ("aln_len", "i8"),
                       ("seed_len", "i8"), ("flags", "u4"), ("_pad", "u4")])
assert PAIR_DTYPE.itemsize == C.sizeof(PairOut)


def default_params(**kw) -> Params:
    p = dict(gamma_tensor=7.0, gamma_coords=0.03, gap_open=1.0, gap_extend=0.01, sw_gap=0.0)
    p.update(kw)
    return Params(**p)


def build(force: bool = False) -> None:
    targets = [HERE / "libcaretta_oracle.so", HERE / "libcaretta_oracle_libm.so"]
    if force or not all(t.exists() for t in targets):
        subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self, libm_exp: bool = False):
        name = "libcaretta_oracle_libm.so" if libm_exp else "libcaretta_oracle.so"
        override = os.environ.get("CARETTA_ORACLE_DIR")          # e.g. a sanitizer build of the same sources
        if override:
            self.lib = lib = C.CDLL(str(Path(override) / name))
        else:
            build()
            self.lib = lib = C.CDLL(str(HERE / name))
        lib.cro_exp.restype = C.c_double
        lib.cro_exp.argtypes = [C.c_double]
        lib.cro_make_score_matrix.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, C.c_int64, C.c_double, _f64p]
        lib.cro_get_rmsd.restype = C.c_double
        lib.cro_get_rmsd.argtypes = [_f64p, _f64p, C.c_int64]
        lib.cro_tm_score.restype = C.c_double
        lib.cro_tm_score.argtypes = [_f64p, _f64p, C.c_int64, C.c_int64, C.c_int64]
        lib.cro_dtw_align.argtypes = [_i64p, C.c_int64, _i64p, C.c_int64, _f64p, C.c_int64, C.c_double, C.c_double,
                                      C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                      C.c_void_p, C.c_void_p]
        lib.cro_smith_waterman_score.restype = C.c_double
        lib.cro_smith_waterman_score.argtypes = [_i64p, C.c_int64, _i64p, C.c_int64, _f64p, C.c_int64, C.c_double]
        lib.cro_smith_waterman.argtypes = [_i64p, C.c_int64, _i64p, C.c_int64, _f64p, C.c_int64, C.c_double,
                                           _i64p, _i64p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        lib.cro_get_common_positions.restype = C.c_int64
        lib.cro_get_common_positions.argtypes = [_i64p, _i64p, C.c_int64, _i64p, _i64p]
        lib.cro_paired_svd_superpose.argtypes = [_f64p, _f64p, C.c_int64, _f64p, _f64p]
        lib.cro_paired_svd_superpose_with_subset.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, _f64p, _f64p,
                                                             C.c_int64, _f64p, _f64p, _f64p]
        lib.cro_apply_rotran.argtypes = [_f64p, C.c_int64, _f64p, _f64p, _f64p]
        lib.cro_svd3.argtypes = [_f64p, _f64p, _f64p, _f64p]
        lib.cro_pairwise_batch.argtypes = [_f64p, _f64p, _i64p, C.c_int64, _i32p, C.c_int64, C.POINTER(Params),
                                           C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        lib.cro_progressive_node_flexible.restype = None
        lib.cro_progressive_node_flexible.argtypes = [_f64p, _f64p, C.c_int64, _f64p, _f64p, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                                      C.c_double, C.c_double, C.c_double, C.c_double, _i64p, _i64p, C.POINTER(C.c_int64),
                                                      _f64p, _f64p]
        lib.cro_progressive_node.restype = C.c_uint32
        lib.cro_progressive_node.argtypes = [_f64p, _f64p, _f64p, C.c_int64, _f64p, _f64p, _f64p, C.c_int64, C.c_int64,
                                             C.c_double, C.c_double, C.POINTER(Params), C.c_double, _i64p, _i64p,
                                             C.POINTER(C.c_int64), _f64p, _f64p, _f64p]
        lib.cro_neighbor_joining.argtypes = [_f64p, C.c_int64, C.c_int, _u64p, _f64p]
        lib.cro_max_threads.restype = C.c_int

    # -- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _f(a):
        return np.ascontiguousarray(a, dtype=np.float64)

    @staticmethod
    def _i(a):
        return np.ascontiguousarray(a, dtype=np.int64)

    def set_sum_mode(self, mode: int):
        self.lib.cro_set_sum_mode(int(mode))

    def exp(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.array([self.lib.cro_exp(float(v)) for v in x.ravel()]).reshape(x.shape)

    def make_score_matrix(self, a, b, gamma):
        a, b = self._f(a), self._f(b)
        s = np.empty((a.shape[0], b.shape[0]))
        self.lib.cro_make_score_matrix(a, a.shape[0], b, b.shape[0], a.shape[1], float(gamma), s)
        return s

    def get_rmsd(self, x1, x2):
        x1, x2 = self._f(x1), self._f(x2)
        return self.lib.cro_get_rmsd(x1, x2, x1.shape[0])

    def tm_score(self, x1, x2, l1, l2):
        x1, x2 = self._f(x1), self._f(x2)
        return self.lib.cro_tm_score(x1, x2, x1.shape[0], int(l1), int(l2))

    def dtw_align(self, seq1, seq2, s, gap_open=0.0, gap_extend=0.0, want_matrices=False):
        seq1, seq2, s = self._i(seq1), self._i(seq2), self._f(s)
        n, m = len(seq1), len(seq2)
        a1 = np.empty(n + m + 1, np.int64)
        a2 = np.empty(n + m + 1, np.int64)
        ln, sc = C.c_int64(0), C.c_double(0)
        mat = bt = None
        mp = bp = None
        if want_matrices:
            mat = np.empty((n + 1, m + 1, 3))
            bt = np.empty((n + 1, m + 1, 3), np.int64)
            mp, bp = mat.ctypes.data, bt.ctypes.data
        self.lib.cro_dtw_align(seq1, n, seq2, m, s, s.shape[1], gap_open, gap_extend, a1.ctypes.data,
                               a2.ctypes.data, C.byref(ln), C.byref(sc), mp, bp)
        out = (a1[:ln.value].copy(), a2[:ln.value].copy(), sc.value)
        return out + (mat, bt) if want_matrices else out

    def dtw_align_score(self, seq1, seq2, s, gap_open=0.0, gap_extend=0.0):
        seq1, seq2, s = self._i(seq1), self._i(seq2), self._f(s)
        sc = C.c_double(0)
        self.lib.cro_dtw_align(seq1, len(seq1), seq2, len(seq2), s, s.shape[1], gap_open, gap_extend, None, None,
                               None, C.byref(sc), None, None)
        return sc.value

    def smith_waterman_score(self, seq1, seq2, s, gap=0.0):
        seq1, seq2, s = self._i(seq1), self._i(seq2), self._f(s)
        return self.lib.cro_smith_waterman_score(seq1, len(seq1), seq2, len(seq2), s, s.shape[1], gap)

    def smith_waterman(self, seq1, seq2, s, gap=0.0):
        seq1, seq2, s = self._i(seq1), self._i(seq2), self._f(s)
        n, m = len(seq1), len(seq2)
        a1 = np.empty(n + m + 1, np.int64)
        a2 = np.empty(n + m + 1, np.int64)
        ln, sc = C.c_int64(0), C.c_double(0)
        rc = self.lib.cro_smith_waterman(seq1, n, seq2, m, s, s.shape[1], gap, a1, a2, C.byref(ln), C.byref(sc))
        return a1[:ln.value].copy(), a2[:ln.value].copy(), sc.value, rc

    def get_common_positions(self, a1, a2):
        a1, a2 = self._i(a1), self._i(a2)
        p1 = np.empty(max(len(a1), 1), np.int64)
        p2 = np.empty(max(len(a1), 1), np.int64)
        k = self.lib.cro_get_common_positions(a1, a2, len(a1), p1, p2)
        return p1[:k].copy(), p2[:k].copy()

    def paired_svd_superpose(self, x1, x2):
        x1, x2 = self._f(x1), self._f(x2)
        r, t = np.empty((3, 3)), np.empty(3)
        self.lib.cro_paired_svd_superpose(x1, x2, x1.shape[0], r, t)
        return r, t

    def paired_svd_superpose_with_subset(self, c1, c2, s1, s2):
        c1, c2, s1, s2 = self._f(c1), self._f(c2), self._f(s1), self._f(s2)
        o1, o2, o3 = np.empty_like(c1), np.empty_like(c2), np.empty_like(s2)
        self.lib.cro_paired_svd_superpose_with_subset(c1, c1.shape[0], c2, c2.shape[0], s1, s2, s1.shape[0], o1, o2, o3)
        return o1, o2, o3

    def apply_rotran(self, x, r, t):
        x = self._f(x)
        out = np.empty_like(x)
        self.lib.cro_apply_rotran(x, x.shape[0], self._f(r), self._f(t), out)
        return out

    def svd3(self, c):
        u, s, vt = np.empty((3, 3)), np.empty(3), np.empty((3, 3))
        self.lib.cro_svd3(self._f(c), u, s, vt)
        return u, s, vt

    def pairwise_batch(self, coords, tensors, offsets, pairs, params: Params | None = None, want_aln=True,
                       nthreads: int = 1):
        """Pipeline H over a pair list.  Returns (structured outs, aln int64 [npairs,2,stride] or None)."""
        coords, tensors, offsets = self._f(coords), self._f(tensors), self._i(offsets)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        params = params or default_params()
        outs = np.zeros(len(pairs), dtype=PAIR_DTYPE)
        lens = np.diff(offsets)
        stride = int(max((lens[pairs[:, 0]] + lens[pairs[:, 1]]).max(), 1)) + 1 if len(pairs) else 1
        aln = np.empty((len(pairs), 2, stride), np.int64) if want_aln else None
        self.lib.cro_pairwise_batch(coords, tensors, offsets, tensors.shape[1], pairs, len(pairs), C.byref(params),
                                    outs.ctypes.data, aln.ctypes.data if want_aln else None, stride, int(nthreads))
        return outs, aln

    def progressive_node(self, x1, t1, w1, x2, t2, w2, mult1, mult2, params: Params | None = None, gamma_weight=1.0):
        """make_intermediate_node for two Proteins -> (aln1, aln2, coords, tensors, weights (len,1), flags)."""
        x1, t1, w1, x2, t2, w2 = (self._f(v) for v in (x1, t1, w1, x2, t2, w2))
        n, m, d = x1.shape[0], x2.shape[0], t1.shape[1]
        params = params or default_params()
        a1, a2 = np.empty(n + m + 1, np.int64), np.empty(n + m + 1, np.int64)
        ln = C.c_int64(0)
        xn, tn, wn = np.empty((n + m, 3)), np.empty((n + m, d)), np.empty(n + m)
        flags = self.lib.cro_progressive_node(x1, t1, w1.reshape(-1), n, x2, t2, w2.reshape(-1), m, d, float(mult1),
                                              float(mult2), C.byref(params), float(gamma_weight), a1, a2, C.byref(ln),
                                              xn, tn, wn)
        k = ln.value
        return a1[:k].copy(), a2[:k].copy(), xn[:k].copy(), tn[:k].copy(), wn[:k].reshape(-1, 1).copy(), flags

    def progressive_node_flexible(self, t1, w1, t2, w2, mult1, mult2, gamma_tensor=0.03, gamma_weight=1.0, gap_open=1.0, gap_extend=0.01):
        """make_intermediate_node with flexible=True in score and mean function -> (aln1, aln2, tensors, weights (len,1))."""
        t1, w1, t2, w2 = (self._f(v) for v in (t1, w1, t2, w2))
        n, m, d = t1.shape[0], t2.shape[0], t1.shape[1]
        a1, a2 = np.empty(n + m + 1, np.int64), np.empty(n + m + 1, np.int64)
        ln = C.c_int64(0)
        tn, wn = np.empty((n + m, d)), np.empty(n + m)
        self.lib.cro_progressive_node_flexible(t1, w1.reshape(-1), n, t2, w2.reshape(-1), m, d, float(mult1), float(mult2), float(gamma_tensor),
                                               float(gamma_weight), float(gap_open), float(gap_extend), a1, a2, C.byref(ln), tn, wn)
        k = ln.value
        return a1[:k].copy(), a2[:k].copy(), tn[:k].copy(), wn[:k].reshape(-1, 1).copy()

    def neighbor_joining(self, d, hoist=True):
        d = self._f(d)
        p = d.shape[0]
        tree = np.zeros((2 * p - 3, 2), np.uint64)
        bl = np.zeros((2 * p - 3, 1))
        n = self.lib.cro_neighbor_joining(d, p, 1 if hoist else 0, tree, bl.reshape(-1))
        assert n == 2 * p - 3
        return tree, bl

    def max_threads(self):
        return self.lib.cro_max_threads()


def tree_bipartitions(tree, num_leaves):
    """Unrooted bipartition set of a caretta NJ tree array (rows are (child, parent))."""
    tree = np.asarray(tree, dtype=np.int64)
    children = {}
    for child, parent in tree:
        children.setdefault(int(parent), []).append(int(child))
    leaves_cache = {}

    def leaves(node):
        if node < num_leaves:
            return frozenset([node])
        if node not in leaves_cache:
            acc = frozenset()
            for c in children.get(node, []):
                acc = acc | leaves(c)
            leaves_cache[node] = acc
        return leaves_cache[node]

    full = frozenset(range(num_leaves))
    parts = set()
    for node in children:
        for c in children[node]:
            side = leaves(c)
            if 1 < len(side) < num_leaves - 1:
                other = full - side
                parts.add(min(side, other, key=lambda s: (len(s), sorted(s))))
    return parts
