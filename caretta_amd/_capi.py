"""ctypes binding of libcaretta_hip (include/caretta_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU
fallback: a missing library or a missing GPU raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libcaretta_hip.so"

CR_NUM_STAGES = 2
FLAG_SEED_SKIPPED, FLAG_METRICS_SKIPPED, FLAG_SEED_ALL_ZERO, FLAG_MEAN_UNSUPERPOSED = 1, 2, 4, 8


CR_ERR_ARGUMENT, CR_ERR_HIP, CR_ERR_MEMORY, CR_ERR_STATE = -1, -2, -3, -4      # include/caretta_hip.h


class CarettaHipError(RuntimeError):
    code = CR_ERR_HIP


class Params(C.Structure):
    _fields_ = [("gamma_tensor", C.c_double), ("gamma_coords", C.c_double), ("gap_open", C.c_double),
                ("gap_extend", C.c_double), ("sw_gap", C.c_double)]


PAIR_RESULT_DTYPE = np.dtype([("sw", "f8"), ("dtw_score", "f8"), ("R", "f8", (9,)), ("t", "f8", (3,)),
                              ("rmsd", "f8"), ("coverage", "f8"), ("tm", "f8"), ("seed_score", "f8"),
                              ("aln_len", "i4"), ("aln_start", "i4"), ("seed_len", "i4"), ("flags", "u4")])
assert PAIR_RESULT_DTYPE.itemsize == 160
# cr_explicit_problem (include/caretta_hip.h)
EXPLICIT_PROBLEM_DTYPE = np.dtype([("s_off", "i8"), ("seq1_off", "i8"), ("seq2_off", "i8"), ("s_rows", "i4"), ("s_cols", "i4"),
                                   ("n", "i4"), ("m", "i4")])
assert EXPLICIT_PROBLEM_DTYPE.itemsize == 40

_vp, _i64, _i32, _f64 = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> argtypes; every entry point returns int (cr_status) except cr_last_error
SIGNATURES = {
    "cr_abi_version": [],
    "cr_device_count": [C.POINTER(C.c_int)],
    "cr_device_trim": [_i32],
    "cr_context_create": [_i32, _vp, _pp],
    "cr_context_create_on_stream": [_i32, _vp, _pp],
    "cr_context_destroy": [_vp],
    "cr_context_synchronize": [_vp],
    "cr_context_stream": [_vp, _pp],
    "cr_context_set_profiling": [_vp, _i32],
    "cr_batch_create": [_vp, _vp, _vp, _vp, _i64, _i64, _pp],
    "cr_batch_set_pairs": [_vp, _vp, _i64],
    "cr_batch_run": [_vp, C.POINTER(Params), _vp],
    "cr_batch_run_scores": [_vp, C.POINTER(Params), _vp],
    "cr_batch_run_tensor_scores": [_vp, C.POINTER(Params), _vp],
    "cr_batch_run_stream_i32": [_vp, C.POINTER(Params), _vp, _vp, _i64, _vp],
    "cr_batch_fetch": [_vp, _vp, _vp, _i64],
    "cr_batch_fetch_i32": [_vp, _vp, _vp, _i64],
    "cr_host_alloc": [C.c_size_t, _pp],
    "cr_host_free": [_vp],
    "cr_batch_fetch_scores": [_vp, _vp, _vp],
    "cr_batch_max_aln_len": [_vp, C.POINTER(C.c_int64)],
    "cr_batch_stage_ms": [_vp, C.POINTER(C.c_float * CR_NUM_STAGES), C.POINTER(C.c_int)],
    "cr_batch_work": [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "cr_batch_layout": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "cr_batch_part_layout": [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)],
    "cr_config_reload": [],
    "cr_plan_layout": [_vp, _i64, _i64, _vp, _i64, _vp, _vp, C.POINTER(C.c_int)],
    "cr_batch_destroy": [_vp],
    "cr_multi_create": [_vp, _i32, _pp],
    "cr_multi_device_count": [_vp, C.POINTER(C.c_int)],
    "cr_multi_pairwise_scores": [_vp, _vp, _vp, _vp, _i64, _i64, C.POINTER(Params), _vp, _vp],
    "cr_multi_last_ms": [_vp, C.POINTER(C.c_float * 3)],
    "cr_multi_numa_nodes": [_vp, _vp],
    "cr_multi_destroy": [_vp],
    "cr_partition_pairs": [_vp, _i64, _i32, _i32, _vp, C.POINTER(C.c_int64)],
    "cr_make_score_matrix": [_vp, _vp, _i64, _vp, _i64, _i64, _f64, _vp],
    "cr_protein_score_function": [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _f64, _f64, _vp, C.POINTER(C.c_uint32)],
    "cr_progressive_node": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _f64, _f64, C.POINTER(Params), _f64,
                            _vp, _vp, C.POINTER(C.c_int64), _vp, _vp, _vp, C.POINTER(C.c_uint32)],
    "cr_progressive_align": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, C.POINTER(Params), _f64, _f64, C.POINTER(_vp)],
    "cr_progressive_align_flexible": [_vp, _vp, _vp, _i64, _i64, _vp, _i64, C.POINTER(Params), _f64, _f64, C.POINTER(_vp)],
    "cr_progressive_sizes": [_vp, _vp],
    "cr_progressive_fetch_msa": [_vp, _vp],
    "cr_progressive_node_table": [_vp, _vp],
    "cr_progressive_fetch_nodes": [_vp, _vp, _vp, _vp, _vp],
    "cr_progressive_destroy": [_vp],
    "cr_dtw_align": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _f64, _f64, _vp, _vp, C.POINTER(C.c_int64),
                     C.POINTER(C.c_double)],
    "cr_smith_waterman_score": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _f64, C.POINTER(C.c_double)],
    "cr_smith_waterman": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _f64, _vp, _vp, C.POINTER(C.c_int64),
                          C.POINTER(C.c_double), C.POINTER(C.c_int)],
    "cr_explicit_batch_create": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _pp],
    "cr_explicit_batch_destroy": [_vp],
    "cr_explicit_batch_last_ms": [_vp, C.POINTER(C.c_float)],
    "cr_smith_waterman_score_batch": [_vp, _f64, _vp],
    "cr_dtw_align_batch": [_vp, _f64, _f64, _vp, _i64, _vp, _vp],
    "cr_smith_waterman_batch": [_vp, _f64, _vp, _i64, _vp, _vp, _vp],
    "cr_paired_svd_superpose": [_vp, _vp, _vp, _i64, _vp, _vp],
    "cr_paired_svd_superpose_with_subset": [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp],
    "cr_apply_rotran": [_vp, _vp, _i64, _vp, _vp, _vp],
    "cr_get_rmsd": [_vp, _vp, _vp, _i64, C.POINTER(C.c_double)],
    "cr_tm_score": [_vp, _vp, _vp, _i64, _i64, _i64, C.POINTER(C.c_double)],
    "cr_msa_metrics": [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp],
    "cr_superpose_core": [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp],
    "cr_superpose_reference": [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp],
    "cr_superpose_members": [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64],
    "cr_mean_axis0": [_vp, _i64, _i64, _vp],
    "cr_get_common_positions": [_vp, _vp, _i64, _vp, _vp, C.POINTER(C.c_int64)],
    "cr_neighbor_joining": [_vp, _i64, _vp, _vp],
    "cr_neighbor_joining_device": [_vp, _vp, _i64, _vp, _vp],
    "cr_assemble_matrix": [_vp, _vp, _i64, _i64, _vp],
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and looks it
    up as "libamdhip64.so"; libcaretta_hip.so needs "libamdhip64.so.7".  If ROCm's copy is loaded first, torch later
    loads its own as well and its runtime finds no GPU ("No HIP GPUs are available").  Loading torch's copy first
    (when a torch installation is present; torch itself is not imported) lets both resolve to the same object.
    CARETTA_SYSTEM_HIP=1 keeps the system runtime."""
    if os.environ.get("CARETTA_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = Path(spec.origin).parent / "lib"
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = libdir / name
        if cand.exists():
            try:
                C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
            except OSError:
                return


def share_torch_rccl():
    """One RCCL per process, for the single-process multi-GPU path (cr_multi_*, which binds librccl at run time and
    prefers a copy that is already loaded).  When a torch installation is present, torch itself is imported here: it then
    loads its librccl.so and that library's dependencies in ITS order.  (Loading torch's librccl.so by path first and
    importing torch later in the same process works until the process exits -- and then aborts in the libraries' teardown
    with "double free or corruption": measured on ROCm 7.0 / torch 2.10, tools/exit_order_probe.py.)  Without torch (or with
    CARETTA_SYSTEM_HIP=1) the library falls back to the loader's search path and /opt/rocm/lib."""
    if os.environ.get("CARETTA_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    if (Path(spec.origin).parent / "lib" / "librccl.so").exists():
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 -- a broken torch installation must not break the one-device paths
            pass


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("CARETTA_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise CarettaHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  caretta_amd has no CPU fallback.")
    _share_torch_hip_runtime()
    lib = C.CDLL(str(path))
    lib.cr_last_error.restype = C.c_char_p
    lib.cr_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = argtypes
    if lib.cr_abi_version() != 1:
        raise CarettaHipError("libcaretta_hip ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int):
    if rc == 0:
        return
    msg = load().cr_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -3:
        raise MemoryError(msg)
    err = CarettaHipError(msg)
    err.code = rc
    raise err


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def i64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)
