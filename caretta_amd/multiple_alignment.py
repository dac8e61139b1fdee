"""Host-side mirror of the pairwise-alignment surface of the reference's
``caretta/multiple_alignment.py``: ``SequenceBase`` (:109-127), ``Protein`` (:312-387),
``MultipleAlignment`` (:148-309), ``tm_score`` (:59-70), ``make_rmsd_coverage_tm_matrix`` (:1000-1055).

The O(P^2) pair loop of ``make_pairwise_matrix`` (:158-170) runs as one batched launch sequence on
the GPU (``engine.PairBatch``); everything numeric goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import typing
from abc import ABC, abstractmethod
from dataclasses import dataclass

import numpy as np

from . import _capi
from . import dynamic_time_warping as dtw
from . import helper
from . import neighbor_joining as nj
from . import score_functions, superposition_functions
from ._capi import check, f64, ptr
from .engine import Context, PairBatch, all_pairs, assemble_matrix, default_context, make_params


def tm_score(coords_1, coords_2, l1, l2):
    """The reference's own TM-like score (multiple_alignment.py:59-70)."""
    x1, x2 = f64(coords_1), f64(coords_2)
    out = C.c_double(0.0)
    check(_capi.load().cr_tm_score(default_context()._h, ptr(x1), ptr(x2), x1.shape[0], int(l1), int(l2),
                                   C.byref(out)))
    return out.value


def _merge_rows(rows_1, rows_2, aln_1, aln_2, combine):
    """One output row per alignment column: a gap on one side takes the other side's row, an aligned column
    `combine(row_1, row_2)` -- the element-wise rule shared by the reference's mean functions."""
    aln_1, aln_2 = np.asarray(aln_1), np.asarray(aln_2)
    has_1, has_2 = aln_1 != -1, aln_2 != -1
    rows_1, rows_2 = np.asarray(rows_1, dtype=np.float64), np.asarray(rows_2, dtype=np.float64)
    out = np.zeros((len(aln_1),) + rows_1.shape[1:])
    both = has_1 & has_2
    out[has_1 & ~has_2] = rows_1[aln_1[has_1 & ~has_2]]
    out[has_2 & ~has_1] = rows_2[aln_2[has_2 & ~has_1]]
    out[both] = combine(rows_1[aln_1[both]], rows_2[aln_2[both]])
    return out


def get_mean_weights(weights_1, weights_2, aln_1, aln_2) -> np.ndarray:
    """multiple_alignment.py:73-82: the consensus weight of a column is the sum of the weights present in it
    ((len, 1) array; 0 + w is w exactly, so the masked form gives the reference's values bit for bit)."""
    w1, w2 = np.asarray(weights_1, dtype=np.float64).reshape(-1, 1), np.asarray(weights_2, dtype=np.float64).reshape(-1, 1)
    return _merge_rows(w1, w2, aln_1, aln_2, lambda a, b: a + b)


class SequenceBase(ABC):
    """Plugin interface of the reference (multiple_alignment.py:109-127)."""
    name: str

    @abstractmethod
    def score_function(self, other: "SequenceBase", **kwargs) -> np.ndarray:
        pass

    def mean_function(self, other: "SequenceBase", aln_1: np.ndarray, aln_2: np.ndarray, name_int: str,
                      **kwargs) -> "SequenceBase":
        pass

    @abstractmethod
    def __len__(self) -> int:
        pass

    @abstractmethod
    def __str__(self) -> str:
        pass


@dataclass
class Protein(SequenceBase):
    name: str
    tensors: np.ndarray
    coordinates: np.ndarray = None
    sequence: str = ""

    def score_function(self, other: "Protein", flexible=False, gamma_tensor=0.03, gamma_coords=0.03,
                       verbose=True) -> np.ndarray:
        """multiple_alignment.py:321-349"""
        if flexible:
            return score_functions.make_score_matrix(self.tensors, other.tensors, score_functions.get_gaussian_score,
                                                     gamma_tensor)
        xi, ti, xj, tj = f64(self.coordinates), f64(self.tensors), f64(other.coordinates), f64(other.tensors)
        if ti.shape[1] != tj.shape[1]:
            raise ValueError("tensor widths differ")
        if ti.shape[1] > MAX_FUSED_TENSOR_WIDTH and (ti.shape[1] > MAX_STAGED_TENSOR_WIDTH or max(ti.shape[0], tj.shape[0]) > _STAGED_MAX_ROWS):
            return self._score_function_wide(xi, ti, xj, tj, other, gamma_tensor, gamma_coords, verbose)
        s = np.empty((xi.shape[0], xj.shape[0]))
        flags = C.c_uint32(0)
        try:
            check(_capi.load().cr_protein_score_function(default_context()._h, ptr(xi), ptr(ti), xi.shape[0], ptr(xj),
                                                         ptr(tj), xj.shape[0], ti.shape[1], float(gamma_tensor),
                                                         float(gamma_coords), ptr(s), C.byref(flags)))
        except ValueError:
            if ti.shape[1] <= MAX_FUSED_TENSOR_WIDTH:
                raise
            return self._score_function_wide(xi, ti, xj, tj, other, gamma_tensor, gamma_coords, verbose)   # (a pair the staged family cannot take)
        if flags.value & _capi.FLAG_SEED_ALL_ZERO:
            raise TypeError("tensor score matrix has no positive local alignment (reference: max_pos is None)")
        if (flags.value & _capi.FLAG_SEED_SKIPPED) and verbose:
            print(f"Too few aligning positions for {self.name} and {other.name}, continuing without superposition")
        return s

    def _score_function_wide(self, xi, ti, xj, tj, other, gamma_tensor, gamma_coords, verbose):
        """multiple_alignment.py:328-349 for tensors wider than the fused kernels are instantiated for (the reference takes any
        (L, d) array, :312-319): the same five steps, each by the drop-in of the function the reference calls there --
        make_score_matrix (any width) -> smith_waterman -> get_common_positions -> paired_svd_superpose_with_subset ->
        make_score_matrix on the superposed frames.  Same values as the fused path computes for d <= 32."""
        tensor_scores = score_functions.make_score_matrix(ti, tj, score_functions.get_gaussian_score, gamma_tensor)
        aln_1, aln_2, _ = dtw.smith_waterman(np.arange(ti.shape[0]), np.arange(tj.shape[0]), tensor_scores)
        pos_1, pos_2 = helper.get_common_positions(aln_1, aln_2)
        if len(pos_1) > 3:
            frame_1, frame_2, _ = superposition_functions.paired_svd_superpose_with_subset(xi, xj, xi[pos_1], xj[pos_2])
        else:
            if verbose:
                print(f"Too few aligning positions for {self.name} and {other.name}, continuing without superposition")
            frame_1, frame_2 = xi, xj
        return score_functions.make_score_matrix(frame_1, frame_2, score_functions.get_gaussian_score, gamma_coords)

    def mean_function(self, other: "Protein", aln_1: np.ndarray, aln_2: np.ndarray, name_int: str, flexible=False,
                      verbose=True) -> "Protein":
        """multiple_alignment.py:351-381: column-wise mean of the tensors, and of the coordinates after superposing
        `other` on `self` over the aligned columns.  (The resident progressive alignment computes the same node on the
        device, cr_progressive.h; this host form serves callers that hold two Proteins and an alignment.)"""
        halve = lambda a, b: (a + b) / 2
        tensors_mean = _merge_rows(self.tensors, other.tensors, aln_1, aln_2, halve)
        if flexible:
            return Protein(name_int, tensors_mean)
        pos_1, pos_2 = helper.get_common_positions(aln_1, aln_2)
        if len(pos_1) > 3:
            frame_1, frame_2, _ = superposition_functions.paired_svd_superpose_with_subset(
                self.coordinates, other.coordinates, self.coordinates[pos_1], other.coordinates[pos_2])
        else:
            if verbose:
                print(f"Too few aligning positions for {self.name} and {other.name}, continuing without superposition")
            frame_1, frame_2 = np.array(self.coordinates), np.array(other.coordinates)
        return Protein(name_int, tensors_mean, _merge_rows(frame_1, frame_2, aln_1, aln_2, halve))

    def __len__(self) -> int:
        return self.tensors.shape[0]

    def __str__(self):
        return self.sequence


def _progressive_node(s1: Protein, s2: Protein, w1, w2, mult1, mult2, name_int, gap_open, gap_extend, gamma_weight,
                      score_function_params, mean_function_params):
    """make_intermediate_node (multiple_alignment.py:193-234) for two Proteins through cr_progressive_node."""
    x1, t1, x2, t2 = f64(s1.coordinates), f64(s1.tensors), f64(s2.coordinates), f64(s2.tensors)
    w1, w2 = f64(w1).reshape(-1), f64(w2).reshape(-1)
    n, m, d = x1.shape[0], x2.shape[0], t1.shape[1]
    if t2.shape[1] != d:
        raise ValueError("tensor widths differ")
    prm = make_params(gamma_tensor=score_function_params.get("gamma_tensor", 0.03),
                      gamma_coords=score_function_params.get("gamma_coords", 0.03),
                      gap_open=gap_open, gap_extend=gap_extend)
    a1, a2 = np.empty(n + m, np.int64), np.empty(n + m, np.int64)
    xn, tn, wn = np.empty((n + m, 3)), np.empty((n + m, d)), np.empty(n + m)
    ln, flags = C.c_int64(0), C.c_uint32(0)
    check(_capi.load().cr_progressive_node(default_context()._h, ptr(x1), ptr(t1), ptr(w1), n, ptr(x2), ptr(t2), ptr(w2),
                                           m, d, float(mult1), float(mult2), C.byref(prm), float(gamma_weight), ptr(a1),
                                           ptr(a2), C.byref(ln), ptr(xn), ptr(tn), ptr(wn), C.byref(flags)))
    if flags.value & _capi.FLAG_SEED_ALL_ZERO:
        raise TypeError("tensor score matrix has no positive local alignment (reference: max_pos is None)")
    verbose = score_function_params.get("verbose", True) or mean_function_params.get("verbose", True)
    if verbose and flags.value & (_capi.FLAG_SEED_SKIPPED | _capi.FLAG_MEAN_UNSUPERPOSED):
        print(f"Too few aligning positions for {s1.name} and {s2.name}, continuing without superposition")
    k = ln.value
    node = Protein(name_int, tn[:k].copy(), xn[:k].copy())
    return a1[:k].copy(), a2[:k].copy(), node, wn[:k].reshape(-1, 1).copy()


_staging = {}          # per thread: page-locked packing buffers of pack_proteins(staging=True), reused from call to call


def pack_proteins(proteins: typing.Sequence[Protein], staging: bool = False):
    """-> (coords (total, 3), tensors (total, d), offsets (P + 1)) float64 / int64, the layout cr_batch_create takes.
    ``staging``: pack into page-locked arrays kept by this module (one set per thread, reused by the next call): the
    upload is then a plain DMA, without the library's own staging copy -- for callers that upload at once and do not
    keep the arrays (make_pairwise_matrix)."""
    lens = [len(p) for p in proteins]
    offsets = np.zeros(len(proteins) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    total, d = int(offsets[-1]), int(np.shape(proteins[0].tensors)[1])
    if staging:
        import threading
        from .engine import pinned_empty
        key = threading.get_ident()
        have = _staging.get(key)
        if have is None or have[0].shape[0] < total or have[1].shape[1] != d:
            grow = max(total, 1) + max(total, 1) // 4
            have = (pinned_empty((grow, 3), np.float64), pinned_empty((grow, d), np.float64))
            _staging[key] = have
        coords, tensors = have[0][:total], have[1][:total]
    else:
        coords, tensors = np.empty((total, 3)), np.empty((total, d))
    if any(p.coordinates is None for p in proteins):      # flexible=True nodes carry tensors only (multiple_alignment.py:361-362)
        coords[:] = 0.0
    else:
        np.concatenate([np.asarray(p.coordinates) for p in proteins], axis=0, out=coords)
    np.concatenate([np.asarray(p.tensors) for p in proteins], axis=0, out=tensors)
    return coords, tensors, offsets


@dataclass
class PairwiseResults:
    """Everything pipeline H produces for a pair list (one entry per pair, in pair order)."""
    pairs: np.ndarray           # (npairs, 2) int32
    results: np.ndarray         # structured, _capi.PAIR_RESULT_DTYPE
    alignments: typing.Optional[np.ndarray]  # (npairs, 2, stride) int64, padded with -2

    def alignment(self, p: int):
        ln = int(self.results["aln_len"][p])
        return self.alignments[p, 0, :ln], self.alignments[p, 1, :ln]


class _NodeAttribute:
    """final_sequences / final_consensus_weights / final_alignments (multiple_alignment.py:248-251).  After a
    device-resident progressive_align the intermediate nodes stay in HBM until one of these is first read."""

    def __init__(self, name):
        self.slot = "_" + name

    def __get__(self, obj, owner):
        if obj is None:
            return self
        if obj.__dict__.get("_pending_nodes") is not None:
            obj._materialize_nodes()
        return obj.__dict__.get(self.slot)

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = value


# Tensor widths (csrc/cr_api.hip padded_width).  The reference takes any (L, d) array (multiple_alignment.py:312-319).
#   d <= 32 : every kernel family (row features in registers, padded to 4 ... 32);
#   d <= 192: the staged family only -- tensor scores by the run-time-width staging kernel, lists of at most 1 024 strips of 64 rows
#             and structures of at most 2 048 residues: `pairwise` hands longer lists over in pieces;
#   wider, or longer structures with d > 32: the same steps through the per-function drop-ins (the plugin route of
#             make_pairwise_matrix / the host walk of progressive_align with Protein._score_function_wide).  Never an error.
MAX_FUSED_TENSOR_WIDTH = 32
MAX_STAGED_TENSOR_WIDTH = 192
_STAGED_MAX_ROWS = 2048

_PLUGIN_BATCH_BYTES = 1 << 30      # score matrices of plugin sequences gathered on the host per batched launch


def _tree_joins(tree):
    """The joins of a neighbor-joining tree in the order progressive_align performs them (multiple_alignment.py:236-246):
    rows come in pairs (child, parent) sharing the parent, the last row joins the two remaining nodes.
    Yields (child_1, child_2, label) with label the parent id, or "final" for the root."""
    tree = np.asarray(tree)
    for row in range(0, tree.shape[0] - 1, 2):
        if int(tree[row + 1, 1]) != int(tree[row, 1]):
            raise AssertionError("rows of the tree do not pair up")
        yield int(tree[row, 0]), int(tree[row + 1, 0]), int(tree[row, 1])
    yield int(tree[-1, 0]), int(tree[-1, 1]), "final"


class _GuideTreeWalk:
    """State of a progressive alignment on the host: the nodes built so far, their consensus weights, and for every
    node the alignment rows of its members (name -> {member name -> index row}), i.e. the reference's
    final_sequences / final_consensus_weights / final_alignments."""

    def __init__(self, leaves, consensus_weight):
        self.nodes = list(leaves)
        self.weights = [np.full((len(s), 1), consensus_weight, dtype=np.float64) for s in leaves]
        self.members = {s.name: {s.name: np.arange(len(s))} for s in leaves}

    def join(self, left, right, label, align_children):
        """Append the parent of nodes `left` and `right`; returns the member rows of the new node."""
        s1, s2 = self.nodes[left], self.nodes[right]
        name_int = f"int-{label}"
        aln_1, aln_2, parent, weights = align_children(s1, s2, self.weights[left], self.weights[right], name_int)
        merged = {}
        for child, aln in ((s1, aln_1), (s2, aln_2)):
            # every member row is re-indexed through the node alignment (:218-229); gaps stay -1
            rows = {name: np.where(aln != -1, np.asarray(row)[aln], -1) for name, row in self.members[child.name].items()}
            self.members[child.name] = rows
            merged.update(rows)
        self.members[name_int] = merged
        self.nodes.append(parent)
        self.weights.append(weights)
        return merged


@dataclass
class MultipleAlignment:
    sequences: typing.List[SequenceBase]
    tree: typing.Optional[np.ndarray] = None
    branch_lengths: typing.Optional[np.ndarray] = None
    alignment: typing.Optional[typing.Dict[str, np.ndarray]] = None
    final_sequences = _NodeAttribute("final_sequences")
    final_consensus_weights = _NodeAttribute("final_consensus_weights")
    final_alignments = _NodeAttribute("final_alignments")

    # -- batched GPU path ---------------------------------------------------------------------
    def _all_proteins(self, need_coordinates: bool = True) -> bool:
        return all(type(s) is Protein and (s.coordinates is not None or not need_coordinates) for s in self.sequences)

    def _fused_width(self, staged_ok: bool = True) -> bool:
        """One tensor width for all sequences, and one the batched engine takes: up to 32 everywhere; up to 192 on staged scores
        (structures of at most 2 048 residues; `staged_ok=False`: callers whose kernels keep the row features in registers)."""
        widths = {np.shape(s.tensors)[1] for s in self.sequences}
        if len(widths) != 1:
            return False
        d = next(iter(widths))
        if d <= MAX_FUSED_TENSOR_WIDTH:
            return True
        return staged_ok and d <= MAX_STAGED_TENSOR_WIDTH and max(len(s) for s in self.sequences) <= _STAGED_MAX_ROWS

    def pairwise(self, score_function_params=None, gap_open_penalty=1.0, gap_extend_penalty=0.01, pairs=None,
                 want_alignments=True, context: typing.Optional[Context] = None,
                 scores_only=False) -> PairwiseResults:
        """Pipeline H over a pair list (default: all i<j): the P x P score entry, the pairwise
        dtw_align alignment and its RMSD / coverage / TM, in one batched launch sequence."""
        prm = dict(score_function_params or {})
        flexible = bool(prm.pop("flexible", False))
        if flexible and not scores_only:
            # (the reference has no pairwise alignment of flexible score matrices either: its 2-sequence branch and its
            # progressive alignment call dtw_align on them, which dynamic_time_warping.dtw_align_batch serves)
            raise ValueError("flexible=True: only the matrix entries (scores_only=True) run on the batched engine")
        prm.pop("verbose", None)
        params = make_params(gamma_tensor=prm.pop("gamma_tensor", 0.03), gamma_coords=prm.pop("gamma_coords", 0.03),
                             gap_open=gap_open_penalty, gap_extend=gap_extend_penalty)
        if prm:
            raise TypeError(f"unknown score_function parameters {sorted(prm)}")
        ctx = context or default_context()
        coords, tensors, offsets = pack_proteins(self.sequences, staging=True)
        pairs = all_pairs(len(self.sequences)) if pairs is None else np.asarray(pairs, np.int32).reshape(-1, 2)
        if tensors.shape[1] > MAX_FUSED_TENSOR_WIDTH and len(pairs):
            return self._pairwise_in_pieces(ctx, coords, tensors, offsets, pairs, params, want_alignments, scores_only)
        batch = PairBatch(ctx, coords, tensors, offsets)
        try:
            batch.set_pairs(pairs)
            batch.run(params, scores_only=scores_only, flexible=flexible)
            if scores_only:                       # the matrix entries only: 12 bytes per pair come back
                sw, flags = batch.fetch_scores()
                # (two columns, not a zeroed PAIR_RESULT record per pair: 21 MB of host memory at 512 structures that nobody
                # reads, 5 ms to fault in, and allocator churn of that size is what used to delay the next kernel --
                # DESIGN.md section 8)
                res, aln = np.zeros(len(pairs), dtype=[("sw", np.float64), ("flags", np.uint32)]), None
                res["sw"], res["flags"] = sw, flags
            else:
                res, aln = batch.fetch(want_alignments)
        finally:
            batch.close()
        if np.any(res["flags"] & _capi.FLAG_SEED_ALL_ZERO):
            bad = pairs[np.nonzero(res["flags"] & _capi.FLAG_SEED_ALL_ZERO)[0][0]]
            raise TypeError(f"tensor score matrix of pair {tuple(bad)} has no positive local alignment "
                            "(reference: max_pos is None)")
        return PairwiseResults(pairs, res, aln)

    def _pairwise_in_pieces(self, ctx, coords, tensors, offsets, pairs, params, want_alignments, scores_only):
        """`pairwise` for tensors wider than 32: the staged family takes at most 1 024 strips of 64 rows (and 2 GiB of staged
        scores) per pair list, so the list goes over in pieces on ONE batch (structures uploaded once); results in pair order."""
        lengths = np.diff(offsets)
        n_max = int(lengths[pairs[:, 0]].max())
        m_max = int(lengths[pairs[:, 1]].max())
        rows_per_lane = max(1, -(-n_max // 512))
        strips = -(-n_max // (64 * rows_per_lane))
        steps = (m_max + 63 + 15) // 16 * 16 + 32
        per_piece = max(1, min(1024 // strips, int((1 << 31) // (8 * strips * steps * rows_per_lane * 64))))
        sw_parts, flag_parts, res_parts, aln_parts = [], [], [], []
        batch = PairBatch(ctx, coords, tensors, offsets)
        try:
            for lo in range(0, len(pairs), per_piece):
                batch.set_pairs(pairs[lo:lo + per_piece])
                batch.run(params, scores_only=scores_only)
                if scores_only:
                    sw, flags = batch.fetch_scores()
                    sw_parts.append(sw.copy())
                    flag_parts.append(flags.copy())
                else:
                    r, a = batch.fetch(want_alignments)
                    res_parts.append(r.copy())
                    aln_parts.append(None if a is None else a.copy())
        finally:
            batch.close()
        if scores_only:
            res, aln = np.zeros(len(pairs), dtype=[("sw", np.float64), ("flags", np.uint32)]), None
            res["sw"], res["flags"] = np.concatenate(sw_parts), np.concatenate(flag_parts)
        else:
            res = np.concatenate(res_parts)
            aln = None
            if want_alignments:
                stride = max(a.shape[2] for a in aln_parts)
                aln = np.full((len(pairs), 2, stride), -2, dtype=aln_parts[0].dtype)
                at = 0
                for a in aln_parts:
                    aln[at:at + len(a), :, :a.shape[2]] = a
                    at += len(a)
        if np.any(res["flags"] & _capi.FLAG_SEED_ALL_ZERO):
            bad = pairs[np.nonzero(res["flags"] & _capi.FLAG_SEED_ALL_ZERO)[0][0]]
            raise TypeError(f"tensor score matrix of pair {tuple(bad)} has no positive local alignment "
                            "(reference: max_pos is None)")
        return PairwiseResults(pairs, res, aln)

    def _pairwise_matrix_multi(self, multi, score_function_params=None):
        """make_pairwise_matrix (multiple_alignment.py:158-170) for Proteins on every device of `multi`
        (engine.MultiDevice): same matrix as the one-device path, bit for bit."""
        prm = dict(score_function_params or {})
        prm.pop("verbose", None)
        prm.pop("flexible", None)
        params = make_params(gamma_tensor=prm.pop("gamma_tensor", 0.03), gamma_coords=prm.pop("gamma_coords", 0.03))
        if prm:
            raise TypeError(f"unknown score_function parameters {sorted(prm)}")
        num = len(self.sequences)
        coords, tensors, offsets = pack_proteins(self.sequences)
        sw, flags = multi.pairwise_scores(coords, tensors, offsets, params)
        pairs = all_pairs(num)
        if np.any(flags & _capi.FLAG_SEED_ALL_ZERO):
            bad = pairs[np.nonzero(flags & _capi.FLAG_SEED_ALL_ZERO)[0][0]]
            raise TypeError(f"tensor score matrix of pair {tuple(bad)} has no positive local alignment "
                            "(reference: max_pos is None)")
        return assemble_matrix(pairs, sw, num)

    def make_pairwise_matrix(self, score_function_params=None):
        """multiple_alignment.py:158-170"""
        if score_function_params is None:
            score_function_params = {}
        num = len(self.sequences)
        if num < 2:
            return np.zeros((num, num))
        if self._all_proteins() and self._fused_width() and not score_function_params.get("flexible", False):
            from . import engine
            npairs = num * (num - 1) // 2
            if npairs >= engine.MULTI_DEVICE_MIN_PAIRS and self._fused_width(staged_ok=False):
                # several GPUs visible to (and owned by) this process: the pair set dealt over all of them, one RCCL
                # all-gather.  A fault of the multi-device machinery (communicator set-up, a device that cannot be opened)
                # is not a fault of the input: say so ONCE, remember it, and compute the same matrix on one device.
                try:
                    multi = engine.multi_device()
                    if multi is not None:
                        return self._pairwise_matrix_multi(multi, score_function_params)
                except _capi.CarettaHipError as exc:
                    import warnings
                    engine.multi_device_failed()
                    warnings.warn(f"multi-GPU pairwise matrix failed ({exc}); running on one device from now on", RuntimeWarning)
            try:
                out = self.pairwise(score_function_params, want_alignments=False, scores_only=True)
                return assemble_matrix(out.pairs, out.results["sw"], num)
            except ValueError:
                # (tensors wider than 32 run on staged scores only: a list even one pair of which does not fit them -- alignment
                # columns beyond the LDS -- takes the plugin route below)
                if np.shape(self.sequences[0].tensors)[1] <= MAX_FUSED_TENSOR_WIDTH:
                    raise
        if (score_function_params.get("flexible", False) and self._all_proteins(need_coordinates=False)
                and self._fused_width(staged_ok=False)):      # (wider tensors: the plugin route below)
            # flexible=True: smith_waterman_score of the tensor score matrix of every pair (multiple_alignment.py:323-326,
            # :164), one launch over the pair list (cr_batch_run_tensor_scores)
            out = self.pairwise(score_function_params, want_alignments=False, scores_only=True)
            return assemble_matrix(out.pairs, out.results["sw"], num)
        # third-party SequenceBase plugins: the score matrices come from the plugin's own score_function (its Python),
        # the O(P^2) smith_waterman_score calls run many matrices per launch (dtw.ExplicitBatch), flushed whenever
        # the matrices held on the host reach _PLUGIN_BATCH_BYTES
        matrix = np.zeros((num, num))
        held, where, held_bytes = [], [], 0

        def flush():
            nonlocal held, where, held_bytes
            if held:
                for (i, j), score in zip(where, dtw.smith_waterman_score_batch(held)):
                    matrix[i, j] = matrix[j, i] = score
            held, where, held_bytes = [], [], 0

        for i in range(num - 1):
            for j in range(i + 1, num):
                scores = np.ascontiguousarray(self.sequences[i].score_function(self.sequences[j], **score_function_params),
                                              dtype=np.float64)
                held.append((np.arange(len(self.sequences[i])), np.arange(len(self.sequences[j])), scores))
                where.append((i, j))
                held_bytes += scores.nbytes
                if held_bytes >= _PLUGIN_BATCH_BYTES:
                    flush()
        flush()
        return matrix

    # -- guide tree + progressive alignment ---------------------------------------------------
    def progressive_align(self, tree, gap_open_penalty, gap_extend_penalty, consensus_weight, gamma_weight,
                          score_function_params=None, mean_function_params=None) -> typing.Dict[str, np.ndarray]:
        """multiple_alignment.py:172-253"""
        mean_function_params = mean_function_params or {}
        score_function_params = score_function_params or {}
        tree = np.asarray(tree)
        flex_score, flex_mean = bool(score_function_params.get("flexible", False)), bool(mean_function_params.get("flexible", False))
        if (len(self.sequences) >= 2 and tree.shape == (2 * len(self.sequences) - 3, 2)
                and all(type(s) is Protein and s.coordinates is not None for s in self.sequences)
                and self._fused_width() and not flex_score and not flex_mean):
            # every node of the tree on the device, one launch pair per tree level (cr_progressive_align)
            wide = np.shape(self.sequences[0].tensors)[1] > MAX_FUSED_TENSOR_WIDTH
            try:
                return self._progressive_align_resident(tree, gap_open_penalty, gap_extend_penalty, consensus_weight,
                                                        gamma_weight, score_function_params, mean_function_params)
            except (_capi.CarettaHipError, ValueError):
                if not wide:
                    raise
                # (tensors wider than 32 run on staged scores only: a tree whose nodes outgrow them takes the host walk below)
        if (len(self.sequences) >= 2 and tree.shape == (2 * len(self.sequences) - 3, 2) and flex_score and flex_mean
                and self._all_proteins(need_coordinates=False) and self._fused_width(staged_ok=False)):
            # flexible=True in score AND mean function (multiple_alignment.py:323-326, :351-362): nodes are tensors and consensus
            # weights only -- the same resident tree without the seed stage (cr_progressive_align_flexible).  A node that outgrows
            # the launch bound sends the tree to the host walk below.
            try:
                return self._progressive_align_resident(tree, gap_open_penalty, gap_extend_penalty, consensus_weight,
                                                        gamma_weight, score_function_params, mean_function_params, flexible=True)
            except _capi.CarettaHipError as exc:
                if exc.code != _capi.CR_ERR_STATE:
                    raise
        self._drop_pending_nodes()
        walk = _GuideTreeWalk(self.sequences, consensus_weight)
        fusable = not score_function_params.get("flexible", False) and not mean_function_params.get("flexible", False)

        def align_children(s1, s2, w1, w2, name_int):
            """One tree node: (aln_1, aln_2, the new consensus sequence, its consensus weights)."""
            size_1, size_2 = len(walk.members[s1.name]), len(walk.members[s2.name])
            mult_1, mult_2 = size_2 / (2 * (size_1 + size_2)), size_1 / (2 * (size_1 + size_2))       # :199-202
            if (fusable and type(s1) is Protein and type(s2) is Protein and s1.coordinates is not None
                    and s2.coordinates is not None and np.shape(s1.tensors)[1] <= MAX_FUSED_TENSOR_WIDTH):      # (cr_progressive_node: registers)
                # the whole node (score matrices, dtw_align, mean_function, get_mean_weights) in two launches
                return _progressive_node(s1, s2, w1, w2, mult_1, mult_2, name_int, gap_open_penalty, gap_extend_penalty,
                                         gamma_weight, score_function_params, mean_function_params)
            # third-party SequenceBase plugins: their score / mean functions around the HIP dtw_align (:204-217)
            scores = s1.score_function(s2, **score_function_params)
            scores += score_functions.make_score_matrix(w1 * mult_1, w2 * mult_2, score_functions.get_gaussian_score,
                                                        gamma_weight)
            aln_1, aln_2, _ = dtw.dtw_align(np.arange(scores.shape[0]), np.arange(scores.shape[1]), scores,
                                            gap_open_penalty=gap_open_penalty, gap_extend_penalty=gap_extend_penalty)
            return (aln_1, aln_2, s1.mean_function(s2, aln_1, aln_2, name_int, **mean_function_params),
                    get_mean_weights(w1, w2, aln_1, aln_2))

        for left, right, label in _tree_joins(tree):
            top = walk.join(left, right, label, align_children)
        self.final_consensus_weights = walk.weights
        self.final_alignments = walk.members
        self.final_sequences = walk.nodes
        return top

    def _progressive_align_resident(self, tree, gap_open_penalty, gap_extend_penalty, consensus_weight, gamma_weight,
                                    score_function_params, mean_function_params, flexible=False):
        """progressive_align (multiple_alignment.py:172-253) for Proteins through cr_progressive_align: the same
        nodes, alignments and attributes, computed level by level of the guide tree with everything resident in HBM."""
        lib = _capi.load()
        P = len(self.sequences)
        # (page-locked packing buffers: cr_progressive_align has uploaded them when it returns)
        coords, tensors, offsets = pack_proteins(self.sequences, staging=True)
        d = tensors.shape[1]
        prm = make_params(gamma_tensor=score_function_params.get("gamma_tensor", 0.03),
                          gamma_coords=score_function_params.get("gamma_coords", 0.03),
                          gap_open=gap_open_penalty, gap_extend=gap_extend_penalty)
        tree_u = np.ascontiguousarray(tree, dtype=np.uint64)
        h = C.c_void_p()
        # the nodes of an earlier call go first: their device blocks (arena, staged scores: hundreds of MB for long
        # structures) return to the library's cache and are handed to this call instead of being allocated beside them
        self._drop_pending_nodes()
        if flexible:
            check(lib.cr_progressive_align_flexible(default_context()._h, ptr(tensors), ptr(offsets), P, d, ptr(tree_u), tree_u.shape[0],
                                                    C.byref(prm), float(consensus_weight), float(gamma_weight), C.byref(h)))
        else:
            check(lib.cr_progressive_align(default_context()._h, ptr(coords), ptr(tensors), ptr(offsets), P, d, ptr(tree_u),
                                           tree_u.shape[0], C.byref(prm), float(consensus_weight), float(gamma_weight),
                                           C.byref(h)))
        try:
            sizes = np.zeros(5, np.int64)
            check(lib.cr_progressive_sizes(h, ptr(sizes)))
            width, num_nodes, total = int(sizes[0]), int(sizes[1]), int(sizes[2])
            table = np.zeros((num_nodes, 6), np.int64)
            check(lib.cr_progressive_node_table(h, ptr(table)))
            msa = np.zeros((P, width), np.int64)
            check(lib.cr_progressive_fetch_msa(h, ptr(msa)))
        except Exception:
            lib.cr_progressive_destroy(h)
            raise
        names = [s.name for s in self.sequences]
        node_names = [f"int-{P + k}" for k in range(num_nodes - 1)] + ["int-final"]
        all_names = names + node_names
        bad = np.flatnonzero(table[:, 4] & _capi.FLAG_SEED_ALL_ZERO)
        if len(bad):
            lib.cr_progressive_destroy(h)
            k = int(bad[0])
            raise TypeError(f"tensor score matrix of {all_names[table[k, 0]]} and {all_names[table[k, 1]]} has no "
                            "positive local alignment (reference: max_pos is None)")
        verbose = score_function_params.get("verbose", True) or mean_function_params.get("verbose", True)
        if verbose:
            for k in np.flatnonzero(table[:, 4] & (_capi.FLAG_SEED_SKIPPED | _capi.FLAG_MEAN_UNSUPERPOSED)):
                print(f"Too few aligning positions for {all_names[table[k, 0]]} and {all_names[table[k, 1]]}, "
                      "continuing without superposition")
        # the alignment in the reference's dict order (:244-247): members of the final node's children, depth first
        order, stack = [], [P + num_nodes - 1]
        while stack:
            x = stack.pop()
            if x < P:
                order.append(x)
            else:
                stack.extend((int(table[x - P, 1]), int(table[x - P, 0])))
        alignment = {names[s]: msa[s] for s in order}
        # the intermediate nodes (:248-251) stay on the device until final_* is read
        self._pending_nodes = dict(handle=h, lib=lib, d=d, total=total, table=table, names=names, node_names=node_names,
                                   consensus_weight=consensus_weight, msa=msa, flexible=flexible)
        self.node_table = table
        return alignment

    def _drop_pending_nodes(self):
        pending = self.__dict__.pop("_pending_nodes", None)
        if pending is not None:
            pending["lib"].cr_progressive_destroy(pending["handle"])

    def __del__(self):
        try:
            self._drop_pending_nodes()
        except Exception:
            pass

    def _materialize_nodes(self):
        """Fetch the intermediate nodes of the last device-resident progressive_align and rebuild
        final_sequences / final_consensus_weights / final_alignments exactly as multiple_alignment.py:193-251 leaves them."""
        pending = self.__dict__.pop("_pending_nodes")
        lib, h, d, total, table = pending["lib"], pending["handle"], pending["d"], pending["total"], pending["table"]
        try:
            aln = np.zeros(2 * total, np.int64)
            flexible = pending.get("flexible", False)
            xn, tn, wn = (None if flexible else np.zeros((total, 3))), np.zeros((total, d)), np.zeros(total)
            check(lib.cr_progressive_fetch_nodes(h, ptr(aln), None if flexible else ptr(xn), ptr(tn), ptr(wn)))
        finally:
            lib.cr_progressive_destroy(h)
        names, node_names, msa = pending["names"], pending["node_names"], pending["msa"]
        P, num_nodes = len(names), len(node_names)
        all_names = names + node_names
        final_sequences = [s for s in self.sequences]
        final_consensus_weights = [np.full((len(s), 1), pending["consensus_weight"], dtype=np.float64) for s in self.sequences]
        members = [[i] for i in range(P)]            # leaf indices below every node, in the reference's dict order
        rows = [np.arange(len(s), dtype=np.int64)[None, :] for s in self.sequences]   # member rows in node columns
        o = 0
        for k in range(num_nodes):
            c1, c2, ln = int(table[k, 0]), int(table[k, 1]), int(table[k, 2])
            a1, a2 = aln[2 * o:2 * o + ln], aln[2 * o + ln:2 * o + 2 * ln]
            # (flexible=True: a node is its mean tensors, multiple_alignment.py:361-362)
            final_sequences.append(Protein(node_names[k], tn[o:o + ln].copy(), None if xn is None else xn[o:o + ln].copy()))
            final_consensus_weights.append(wn[o:o + ln].reshape(-1, 1).copy())
            rows[c1] = np.where(a1 != -1, rows[c1][:, a1], -1)          # :218-229, all member rows at once
            rows[c2] = np.where(a2 != -1, rows[c2][:, a2], -1)
            members.append(members[c1] + members[c2])
            rows.append(np.vstack([rows[c1], rows[c2]]))
            o += ln
        root = P + num_nodes - 1
        assert all(np.array_equal(rows[root][r], msa[s]) for r, s in enumerate(members[root]))
        self.final_alignments = {all_names[x]: {names[s]: rows[x][r] for r, s in enumerate(members[x])}
                                 for x in range(len(all_names))}
        self.final_consensus_weights = final_consensus_weights
        self.final_sequences = final_sequences

    def multiple_align(self, pairwise_distance_matrix, gap_open_penalty, gap_extend_penalty, consensus_weight,
                       gamma_weight, score_function_params=None, mean_function_params=None):
        """multiple_alignment.py:255-285"""
        mean_function_params = mean_function_params or {}
        score_function_params = score_function_params or {}
        if (len(self.sequences) == 2 and self._all_proteins() and self._fused_width()
                and not score_function_params.get("flexible", False)):
            # :263-275 is pipeline H of this one pair: both score matrices, the seed and dtw_align in two launches
            out = self.pairwise(score_function_params, gap_open_penalty, gap_extend_penalty)
            if score_function_params.get("verbose", True) and out.results["flags"][0] & _capi.FLAG_SEED_SKIPPED:
                print(f"Too few aligning positions for {self.sequences[0].name} and {self.sequences[1].name}, "
                      "continuing without superposition")
            aln_1, aln_2 = out.alignment(0)
            self.alignment = {self.sequences[0].name: aln_1.copy(), self.sequences[1].name: aln_2.copy()}
            return self.alignment
        if len(self.sequences) == 2:
            score_matrix = self.sequences[0].score_function(self.sequences[1], **score_function_params)
            aln_1, aln_2, _ = dtw.dtw_align(np.arange(score_matrix.shape[0]), np.arange(score_matrix.shape[1]),
                                            score_matrix, gap_open_penalty=gap_open_penalty,
                                            gap_extend_penalty=gap_extend_penalty)
            self.alignment = {self.sequences[0].name: aln_1, self.sequences[1].name: aln_2}
            return self.alignment
        self.tree, self.branch_lengths = nj.neighbor_joining(pairwise_distance_matrix)
        self.alignment = self.progressive_align(self.tree, gap_open_penalty, gap_extend_penalty, consensus_weight,
                                                gamma_weight, score_function_params, mean_function_params)
        return self.alignment

    def to_sequence_alignment(self, alignment=None):
        """multiple_alignment.py:287-297"""
        alignment = self.alignment if alignment is None else alignment
        out = {}
        for p in self.sequences:
            sequence = str(p)
            out[p.name] = "".join(sequence[i] if i != -1 else "-" for i in alignment[p.name])
        return out

    def write_alignment(self, fasta_file, alignment=None):
        """multiple_alignment.py:299-309"""
        alignment = self.alignment if alignment is None else alignment
        with open(fasta_file, "w") as f:
            for p in self.sequences:
                sequence = str(p)
                aligned = "".join(sequence[i] if i != -1 else "-" for i in alignment[p.name])
                f.write(f">{p.name}\n{aligned}\n")


def alignment_to_numpy(alignment: typing.Dict[str, str]) -> typing.Dict[str, np.ndarray]:
    """Gapped sequences ("AC-D") -> index arrays ([0, 1, -1, 2]) (multiple_alignment.py:30-42)."""
    out = {}
    for name, gapped in alignment.items():
        present = np.frombuffer(str(gapped).encode("ascii", "replace"), dtype=np.uint8) != ord("-")
        out[name] = np.where(present, np.cumsum(present) - 1, -1)
    return out


def trigger_numba_compilation():
    """The reference warms up its njit functions here (multiple_alignment.py:1058-1076).  Nothing is compiled at run
    time in this package; the call loads libcaretta_hip.so and creates the device context, so that the first
    alignment does not pay for it."""
    default_context()


def make_rmsd_coverage_tm_matrix(alignment, proteins, superpose_first: bool = True):
    """multiple_alignment.py:1000-1055 (see caretta_amd.msa_superposition)."""
    from .msa_superposition import make_rmsd_coverage_tm_matrix as impl
    return impl(alignment, proteins, superpose_first)


def __getattr__(name):
    # superpose*, get_reference_structures, make_coverage_gap_distance_matrix live in msa_superposition
    from . import msa_superposition
    if hasattr(msa_superposition, name):
        return getattr(msa_superposition, name)
    raise AttributeError(name)
