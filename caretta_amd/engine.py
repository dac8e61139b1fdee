"""Host-side handles over the C ABI: a device context and a resident batch of structures.

``Context``  = device + HIP stream (optionally borrowed from torch) + per-stage profiling events.
``PairBatch`` = packed structures resident in HBM plus the scratch of one pair list; ``run()`` enqueues
the four stages of the pairwise pipeline, ``fetch()`` brings results to the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _capi
from ._capi import Params, check, f64, i64, ptr

DEFAULT_PARAMS = dict(gamma_tensor=7.0, gamma_coords=0.03, gap_open=1.0, gap_extend=0.01, sw_gap=0.0)


def make_params(**kw) -> Params:
    p = dict(DEFAULT_PARAMS)
    for k, v in kw.items():
        if k not in p:
            raise TypeError(f"unknown parameter {k!r}")
        p[k] = float(v)
    return Params(**p)


def device_count() -> int:
    n = C.c_int(0)
    rc = _capi.load().cr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Context:
    def __init__(self, device: int = 0, stream: Optional[int] = None, profiling: int = 0):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        if stream is None:       # a private (blocking) stream
            check(self._lib.cr_context_create(int(device), None, C.byref(self._h)))
        else:                    # borrow the caller's stream; 0 is the legacy default stream torch works on
            check(self._lib.cr_context_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(self._h)))
        self.device = int(device)
        if profiling:
            self.set_profiling(profiling)

    def set_profiling(self, slots: int):
        """Record per-stage HIP events for the next ``slots`` runs (ring); 0 switches it off."""
        check(self._lib.cr_context_set_profiling(self._h, int(slots)))

    def synchronize(self):
        check(self._lib.cr_context_synchronize(self._h))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        check(self._lib.cr_context_stream(self._h, C.byref(s)))
        return s.value or 0

    def close(self):
        if self._h:
            self._lib.cr_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class _PinnedBlock:
    """Owner of one cr_host_alloc block (freed when the last array built on it is gone)."""

    def __init__(self, nbytes: int):
        self._lib = _capi.load()
        self.ptr = C.c_void_p()
        check(self._lib.cr_host_alloc(max(int(nbytes), 1), C.byref(self.ptr)))

    def __del__(self):
        try:
            if self.ptr:
                self._lib.cr_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """numpy array in page-locked host memory (device copies into it run at DMA speed)."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dtype.itemsize, 1)
    block = _PinnedBlock(nbytes)
    buf = (C.c_char * nbytes).from_address(block.ptr.value)
    buf._owner = block           # the array's base chain holds `buf`, `buf` holds the block
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def all_pairs(num: int) -> np.ndarray:
    """(i, j), i < j in the row-major order of multiple_alignment.py:162-163."""
    i, j = np.triu_indices(num, k=1)
    return np.ascontiguousarray(np.stack([i, j], axis=1), dtype=np.int32)


LAYOUT_NAMES = ("single", "team", "wide", "staged", "duo", "trio", "classes")


def reload_config():
    """Have the library read its calibration switches (CARETTA_* environment variables, caretta_amd/csrc/cr_config.h) again:
    they are read once when the library is loaded.  Measurement tools and tests that change os.environ call this."""
    check(_capi.load().cr_config_reload())


def plan_layout(offsets, d, pairs):
    """What ``PairBatch.set_pairs`` would decide for this pair list, without a device (cr_plan_layout): ->
    ([(kernel family, rows per lane A, rows per lane B, strips with A, pairs)] per part, part index of every pair)."""
    offsets, pairs = i64(offsets), np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    parts, cls, n = np.zeros((3, 5), np.int32), np.zeros(len(pairs), np.int32), C.c_int(0)
    check(_capi.load().cr_plan_layout(ptr(offsets), len(offsets) - 1, int(d), ptr(pairs), len(pairs), ptr(cls), ptr(parts), C.byref(n)))
    return [(LAYOUT_NAMES[int(r[0])], int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in parts[:n.value]], cls


class PairBatch:
    def __init__(self, ctx: Context, coords, tensors, offsets):
        self.ctx = ctx
        self._lib = ctx._lib
        self.coords, self.tensors, self.offsets = f64(coords), f64(tensors), i64(offsets)
        if self.coords.ndim != 2 or self.coords.shape[1] != 3:
            raise ValueError("coords must have shape (total, 3)")
        if self.tensors.ndim != 2 or self.tensors.shape[0] != self.coords.shape[0]:
            raise ValueError("tensors must have shape (total, d)")
        if self.offsets.ndim != 1 or self.offsets[-1] != self.coords.shape[0]:
            raise ValueError("offsets must end at the total residue count")
        self.num_structures = len(self.offsets) - 1
        self._h = C.c_void_p()
        check(self._lib.cr_batch_create(ctx._h, ptr(self.coords), ptr(self.tensors), ptr(self.offsets),
                                        self.num_structures, self.tensors.shape[1], C.byref(self._h)))
        self.pairs = np.zeros((0, 2), np.int32)

    def set_pairs(self, pairs: Sequence) -> "PairBatch":
        self.pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        check(self._lib.cr_batch_set_pairs(self._h, ptr(self.pairs), len(self.pairs)))
        return self

    def run(self, params: Optional[Params] = None, sw_out_device_ptr: Optional[int] = None, scores_only: bool = False,
            flexible: bool = False):
        """Enqueue the pipeline.  ``scores_only``: just the P x P matrix entries (make_pairwise_matrix): the seed kernel and
        a smith_waterman_score kernel, no pairwise dtw_align -- afterwards only ``fetch_scores`` has results.
        ``flexible`` (with ``scores_only``): the entries of the flexible=True score function, smith_waterman_score of the
        tensor score matrix alone (cr_batch_run_tensor_scores)."""
        params = params or make_params()
        if flexible and not scores_only:
            raise ValueError("flexible=True has no pairwise alignment stage: use scores_only=True")
        fn = self._lib.cr_batch_run_tensor_scores if flexible else self._lib.cr_batch_run_scores if scores_only else self._lib.cr_batch_run
        check(fn(self._h, C.byref(params), C.c_void_p(sw_out_device_ptr) if sw_out_device_ptr else None))

    def run_streamed(self, params: Optional[Params] = None, want_alignments: bool = True, sw_out_device_ptr: Optional[int] = None):
        """``run`` with the download folded in (cr_batch_run_stream_i32): the alignment kernel writes every pair's rows
        (int32) and record straight into page-locked arrays kept by this batch.  Asynchronous; -> (results, aln) that are
        complete after ``ctx.synchronize()`` and are REUSED by the next call.  Only the first ``results["aln_len"][p]`` entries
        of each row of pair p are written: what lies behind them is whatever an earlier batch (or nothing) left there --
        ``fetch`` pads the same cells with -2, this call does not touch them."""
        params = params or make_params()
        n = len(self.pairs)
        stride = 0
        if want_alignments:
            mx = C.c_int64(0)
            check(self._lib.cr_batch_max_aln_len(self._h, C.byref(mx)))
            stride = max(int(mx.value), 1)
        res = self._pinned("res", (n,), _capi.PAIR_RESULT_DTYPE)
        aln = self._pinned("aln", (n, 2, stride), np.int32) if want_alignments else None
        check(self._lib.cr_batch_run_stream_i32(self._h, C.byref(params), ptr(res), ptr(aln) if aln is not None else None, stride,
                                                C.c_void_p(sw_out_device_ptr) if sw_out_device_ptr else None))
        return res, aln

    def fetch(self, want_alignments: bool = True, pinned: bool = False):
        """-> (structured array of per-pair results, aln [npairs, 2, stride] or None).

        Default: fresh numpy arrays, int64 rows as the reference's.  ``pinned=True``: int32 rows in page-locked arrays that
        this batch keeps and REUSES for the next fetch (DMA-speed copies; copy what has to outlive the next fetch)."""
        n = len(self.pairs)
        stride = 0
        if want_alignments:
            mx = C.c_int64(0)
            check(self._lib.cr_batch_max_aln_len(self._h, C.byref(mx)))
            stride = max(int(mx.value), 1)
        if pinned:
            res = self._pinned("res", (n,), _capi.PAIR_RESULT_DTYPE)
            aln = self._pinned("aln", (n, 2, stride), np.int32) if want_alignments else None
            check(self._lib.cr_batch_fetch_i32(self._h, ptr(res), ptr(aln) if aln is not None else None, stride))
            return res, aln
        res = np.zeros(n, dtype=_capi.PAIR_RESULT_DTYPE)
        aln = np.empty((n, 2, stride), dtype=np.int64) if want_alignments else None
        check(self._lib.cr_batch_fetch(self._h, ptr(res), ptr(aln) if aln is not None else None, stride))
        return res, aln

    def _pinned(self, key, shape, dtype):
        """A page-locked numpy array kept by this batch (allocated once per shape)."""
        cache = self.__dict__.setdefault("_pinned_cache", {})
        have = cache.get(key)
        if have is not None and have.shape == tuple(shape) and have.dtype == np.dtype(dtype):
            return have
        arr = pinned_empty(shape, dtype)
        cache[key] = arr
        return arr

    def fetch_scores(self):
        """-> (sw f64[npairs], flags u32[npairs]): the P x P matrix entries only (8 + 4 bytes per pair)."""
        n = len(self.pairs)
        sw, flags = np.zeros(n), np.zeros(n, np.uint32)
        check(self._lib.cr_batch_fetch_scores(self._h, ptr(sw), ptr(flags)))
        return sw, flags

    def stage_ms(self):
        """(per-stage device ms averaged over the recorded runs, number of runs averaged)."""
        buf = (C.c_float * _capi.CR_NUM_STAGES)()
        n = C.c_int(0)
        check(self._lib.cr_batch_stage_ms(self._h, C.byref(buf), C.byref(n)))
        return np.array(list(buf), dtype=np.float64), n.value

    def work(self):
        """(algorithmic HBM bytes, DP cells per pass) of one run of the current pair list."""
        b, c = C.c_double(0), C.c_double(0)
        check(self._lib.cr_batch_work(self._h, C.byref(b), C.byref(c)))
        return b.value, c.value

    def layout(self):
        """(kernel family, rows per lane A, rows per lane B, strips with A) chosen by ``set_pairs`` -- "single" (one wave per
        pair), "team", "wide" (one workgroup per pair), "staged" (scores by their own launches), "duo" / "trio" (mid-size lists: split by rows / by function)."""
        f, ra, rb, na = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(self._lib.cr_batch_layout(self._h, C.byref(f), C.byref(ra), C.byref(rb), C.byref(na)))
        return LAYOUT_NAMES[f.value], ra.value, rb.value, na.value

    def part_layouts(self):
        """[(kernel family, rows per lane A, rows per lane B, strips with A, pairs)] of the size classes a ragged list was split
        into (``layout()[0] == "classes"``), or of the one list."""
        count = self.layout()[1] if self.layout()[0] == "classes" else 1
        out = []
        for k in range(count):
            f, ra, rb, na, n = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int64(0)
            check(self._lib.cr_batch_part_layout(self._h, k, C.byref(f), C.byref(ra), C.byref(rb), C.byref(na), C.byref(n)))
            out.append((LAYOUT_NAMES[f.value], ra.value, rb.value, na.value, n.value))
        return out

    def close(self):
        if self._h:
            self._lib.cr_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDevice:
    """Several GPUs of one node driven from THIS process (cr_multi_*): one context per device, the pair set dealt over
    them, one grouped RCCL all-gather of the score vectors.  ``devices``: indices, or None for every visible device."""

    def __init__(self, devices: Optional[Sequence[int]] = None):
        self._lib = _capi.load()
        _capi.share_torch_rccl()
        self._h = C.c_void_p()
        if devices is None:
            check(self._lib.cr_multi_create(None, 0, C.byref(self._h)))
        else:
            arr = np.ascontiguousarray(devices, dtype=np.int32)
            check(self._lib.cr_multi_create(ptr(arr), len(arr), C.byref(self._h)))
        n = C.c_int(0)
        check(self._lib.cr_multi_device_count(self._h, C.byref(n)))
        self.num_devices = n.value

    def pairwise_scores(self, coords, tensors, offsets, params: Optional[Params] = None):
        """-> (sw f64[npairs], flags u32[npairs]) for all pairs i < j in row-major order (``all_pairs``)."""
        coords, tensors, offsets = f64(coords), f64(tensors), i64(offsets)
        num = len(offsets) - 1
        npairs = num * (num - 1) // 2
        params = params or make_params()
        sw, flags = np.empty(npairs), np.zeros(npairs, np.uint32)
        check(self._lib.cr_multi_pairwise_scores(self._h, ptr(coords), ptr(tensors), ptr(offsets), num, tensors.shape[1],
                                                 C.byref(params), ptr(sw), ptr(flags)))
        return sw, flags

    def last_ms(self):
        """ms of the last call's phases: (slowest device's upload + kernels, all-gather, download + scatter to pair order) -- the
        first two and the download from events on the devices' streams, the scatter as host time"""
        buf = (C.c_float * 3)()
        check(self._lib.cr_multi_last_ms(self._h, C.byref(buf)))
        return tuple(buf)

    def numa_nodes(self):
        """Per device: the NUMA node its host thread was pinned to (CARETTA_MULTI_NUMA=1), else -1."""
        nodes = np.full(self.num_devices, -1, dtype=np.int32)
        check(self._lib.cr_multi_numa_nodes(self._h, ptr(nodes)))
        return [int(x) for x in nodes]

    def close(self):
        if self._h:
            self._lib.cr_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_multi = None
_MULTI_FAILED = object()              # sentinel: the multi-device machinery could not be set up in this process
MULTI_DEVICE_MIN_PAIRS = 16384        # below this one GPU is not even full (256 CUs x 16 wave slots = 4096 pairs per round)


def _inside_multi_rank_job() -> bool:
    """One process per GPU (torchrun, bench.py's own ranks, any torch.distributed job): every rank sees every GPU, and a
    MultiDevice in each of them would open WORLD_SIZE x WORLD_SIZE contexts and communicators on the librccl shared with torch."""
    import os
    import sys
    try:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            return True
    except ValueError:
        pass
    dist = getattr(sys.modules.get("torch"), "distributed", None)      # never imports torch by itself
    try:
        return bool(dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
    except Exception:
        return False


def multi_device() -> Optional[MultiDevice]:
    """The process-wide MultiDevice over all visible GPUs, or None: one GPU only, CARETTA_SINGLE_DEVICE=1, this process is one
    rank of a multi-rank job (it owns ONE of the GPUs it sees; CARETTA_MULTI_DEVICE=1 overrides), or setting it up has failed
    before in this process (``multi_device_failed`` records that; the caller computes on one device)."""
    global _multi
    import os
    if os.environ.get("CARETTA_SINGLE_DEVICE") == "1" or _multi is _MULTI_FAILED:
        return None
    if _inside_multi_rank_job() and os.environ.get("CARETTA_MULTI_DEVICE") != "1":
        return None
    if _multi is None:
        if device_count() < 2:
            return None
        _multi = MultiDevice()
    return _multi


def multi_device_failed() -> None:
    """Remember that the multi-device path does not work here (no usable RCCL, a device that cannot be opened): later
    calls of ``multi_device`` return None at once instead of computing every share and failing again."""
    global _multi
    old, _multi = _multi, _MULTI_FAILED
    if isinstance(old, MultiDevice):
        try:
            old.close()
        except Exception:
            pass


def partition_pairs(lengths, world: int, rank: int) -> np.ndarray:
    """cr_partition_pairs: indices into ``all_pairs(len(lengths))`` owned by ``rank`` of ``world`` (host only)."""
    lengths = i64(lengths)
    num = len(lengths)
    cap = (num * (num - 1) // 2 + world - 1) // world
    idx = np.zeros(max(cap, 1), np.int64)
    cnt = C.c_int64(0)
    check(_capi.load().cr_partition_pairs(ptr(lengths), num, int(world), int(rank), ptr(idx), C.byref(cnt)))
    return idx[:cnt.value].copy()


def assemble_matrix(pairs, scores, num: int) -> np.ndarray:
    """Symmetric P x P score matrix with a zero diagonal (multiple_alignment.py:161-169)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    scores = f64(scores)
    out = np.zeros((num, num))
    check(_capi.load().cr_assemble_matrix(ptr(pairs), ptr(scores), len(pairs), num, ptr(out)))
    return out
