"""What bench.py is made of: the rank's runtime, one configuration's pair set sharded over the ranks, the correctness gates
(oracle, N ranks against one GPU) and the measurement programmes behind the headline.  bench.py keeps the contract (argument
parsing, the timed region, the JSON line); everything here is imported by it and by the tests.

The world-size-2 gloo test (tests/test_distributed_cpu.py) drives `Sharded`, `multi_gpu_record` and `rank_records` with a
runtime whose compute is injected (`CpuRuntime`): the deal, the all-gather, the gates and the exit code are the code the
N-GPU line runs.
"""
from __future__ import annotations

import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64, spec; counts an FMA as 2
CONFIGS = {"c2": (32, 150, 20241), "c3": (128, 300, 20242), "c4": (512, 300, 20243), "c5": (64, 1200, 20244)}


def workload(name: str):
    """(structures, residues, seed).  `headline` IS BASELINE config 3 -- 128 x 300, the pair set FIXED at every N (strong
    scaling, as the north star words it: ">= 6x further scaling at 8 GPUs" of the 128 x 300 job)."""
    return CONFIGS["c3" if name == "headline" else name]


def weak_workload(n_gpus: int):
    """The weak-scaling companion of the headline: P = round(128 sqrt(N)) structures of 300, per-GPU work fixed."""
    return int(round(128 * math.sqrt(n_gpus))), 300, 20242


def stage_bytes(lengths, pairs, d):
    """Algorithmic HBM bytes per launch of the two fill kernels, SURVEY.md 8(d) / DESIGN.md section 5:
    k_seed reads the two structures' tensors and writes 2 bits per cell (70 500 B per 300 x 300 pair, d = 10); k_align reads the
    coordinates and writes 4 bits per cell, the two alignment rows and the pair's record (69 136 B).  `*_readback` adds the
    traceback's re-read of the decision words and the small per-pair records (the round-1 figure)."""
    n = lengths[pairs[:, 0]].astype(np.float64)
    m = lengths[pairs[:, 1]].astype(np.float64)
    seed = 8.0 * d * (n + m) + n * m / 4
    align = 24.0 * (n + m) + n * m / 2 + 16.0 * (n + m) + 136.0
    seed_rb = seed + n * m / 4 + 24.0 * (n + m) + 144
    align_rb = align + n * m / 2 + 144 + 160
    return {"k_seed": float(seed.sum()), "k_align": float(align.sum()),
            "k_seed_readback": float(seed_rb.sum()), "k_align_readback": float(align_rb.sum())}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ----------------------------------------------------------------------------------------------------------------------
# the rank's runtime
# ----------------------------------------------------------------------------------------------------------------------
class GpuRuntime:
    """One rank of bench.py on its MI355X: torch device + stream, the library context on that stream, the RCCL group."""

    def __init__(self, world, rank, local_rank, use_dist, dim=10):
        import torch
        import torch.distributed as dist
        from caretta_amd import engine
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.local_rank, self.use_dist, self.dim = world, rank, local_rank, use_dist, dim
        self.dev = torch.device("cuda", local_rank)
        # the kernels run on torch's current stream (the legacy default stream, handle 0, unless the caller changed it):
        # the all-gather that follows cr_batch_run is ordered behind the kernels by the stream itself
        self.ctx = engine.Context(local_rank, stream=torch.cuda.current_stream(self.dev).cuda_stream)
        self.params = engine.make_params()

    # -- process group
    def fence(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds: float) -> float:
        if not self.use_dist:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather(self, out, local):
        self.dist.all_gather_into_tensor(out, local)

    def all_gather_object(self, obj):
        if not self.use_dist:
            return [obj]
        got = [None] * self.world
        self.dist.all_gather_object(got, obj)
        return got

    def broadcast_flag(self, flag: bool) -> bool:
        """rank 0's verdict to every rank (so that every rank exits with the same code)"""
        if not self.use_dist:
            return flag
        t = self.torch.tensor([1.0 if flag else 0.0], dtype=self.torch.float64, device=self.dev)
        self.dist.broadcast(t, src=0)
        return bool(t.item() != 0.0)

    # -- device memory and compute
    def new_local(self, count):
        return self.torch.full((count,), float("nan"), dtype=self.torch.float64, device=self.dev)

    def new_gathered(self, count):
        return self.torch.empty(count, dtype=self.torch.float64, device=self.dev)

    def make_batch(self, coords, tensors, offsets, pairs):
        from caretta_amd import engine
        return engine.PairBatch(self.ctx, coords, tensors, offsets).set_pairs(pairs)

    def run_batch(self, batch, local, scores_only):
        batch.run(self.params, sw_out_device_ptr=local.data_ptr(), scores_only=scores_only)

    def timed_on_stream(self, fn) -> float:
        """milliseconds `fn`'s device work takes, from events on the stream it is queued on (torch's current stream: the
        kernels and the all-gather both run there)"""
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        return float(e0.elapsed_time(e1))

    def to_numpy(self, t):
        return t.cpu().numpy()

    # -- who ran where
    def rank_record(self):
        p = self.torch.cuda.get_device_properties(self.dev)
        rec = {"rank": self.rank, "local_rank": self.local_rank, "device": int(self.dev.index), "name": p.name,
               "gcn_arch": getattr(p, "gcnArchName", None), "pid": os.getpid(), "host": os.uname().nodename}
        try:
            rec["pci_bus_id"] = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except AttributeError:
            rec["pci_bus_id"] = None
        try:
            rec["uuid"] = str(p.uuid)
        except AttributeError:
            pass
        return rec

    def collective_library(self):
        out = {"backend": self.dist.get_backend() if self.use_dist else None}
        try:
            v = self.torch.cuda.nccl.version()
            out["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception as exc:                              # noqa: BLE001
            out["rccl_version"] = f"unavailable ({exc!r})"
        out["hip"] = getattr(self.torch.version, "hip", None)
        return out


class CpuRuntime:
    """The same interface on CPU tensors over gloo with the per-pair compute INJECTED (`compute(coords, tensors, offsets,
    pairs) -> sw scores`): what the world-size-2 test runs the sharding, the gather and the gates on.  Never used by bench.py."""

    def __init__(self, world, rank, compute, use_dist=None, dim=10):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank, self.local_rank, self.dim = world, rank, rank, dim
        self.use_dist = (world > 1) if use_dist is None else use_dist
        self.compute = compute
        self.params = None

    def fence(self):
        if self.use_dist:
            self.dist.barrier()

    def sync(self):
        pass

    def max_over_ranks(self, seconds):
        if not self.use_dist:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather(self, out, local):
        self.dist.all_gather_into_tensor(out, local)

    def all_gather_object(self, obj):
        if not self.use_dist:
            return [obj]
        got = [None] * self.world
        self.dist.all_gather_object(got, obj)
        return got

    def broadcast_flag(self, flag):
        if not self.use_dist:
            return flag
        t = self.torch.tensor([1.0 if flag else 0.0], dtype=self.torch.float64)
        self.dist.broadcast(t, src=0)
        return bool(t.item() != 0.0)

    def new_local(self, count):
        return self.torch.full((count,), float("nan"), dtype=self.torch.float64)

    def new_gathered(self, count):
        return self.torch.empty(count, dtype=self.torch.float64)

    def make_batch(self, coords, tensors, offsets, pairs):
        return _InjectedBatch(self.compute, coords, tensors, offsets, pairs)

    def run_batch(self, batch, local, scores_only):
        local[:len(batch.pairs)] = self.torch.from_numpy(np.asarray(batch.scores(), dtype=np.float64))

    def timed_on_stream(self, fn):
        t0 = time.perf_counter()
        fn()
        return (time.perf_counter() - t0) * 1e3

    def to_numpy(self, t):
        return t.numpy()

    def rank_record(self):
        return {"rank": self.rank, "local_rank": self.rank, "device": None, "name": "cpu (injected compute)", "pid": os.getpid(),
                "host": os.uname().nodename, "pci_bus_id": None}

    def collective_library(self):
        return {"backend": self.dist.get_backend() if self.use_dist else None, "rccl_version": None}


class _InjectedBatch:
    def __init__(self, compute, coords, tensors, offsets, pairs):
        self.compute, self.coords, self.tensors, self.offsets, self.pairs = compute, coords, tensors, offsets, np.asarray(pairs)
        self._scores = None

    def scores(self):
        if self._scores is None:
            self._scores = self.compute(self.coords, self.tensors, self.offsets, self.pairs)
        return self._scores

    def layout(self):
        return ("injected",)

    def close(self):
        pass


# ----------------------------------------------------------------------------------------------------------------------
# one configuration, its pair set sharded over the ranks
# ----------------------------------------------------------------------------------------------------------------------
class Sharded:
    """One config's pair set sharded over the ranks: the batch on this rank's share + the all-gather buffers.
    ranks / me: the deal (default: the process group's); stride: every stride-th pair on this one rank (one GPU's share of a
    split, run alone)."""

    def __init__(self, rt, num, length, seed, ranks=None, me=None, stride=None, family=None):
        from caretta_amd import distributed as cdist
        from caretta_amd import engine, synthetic
        self.rt = rt
        ranks = rt.world if ranks is None else ranks
        me = rt.rank if me is None else me
        self.num, self.length, self.seed = num, length, seed
        fam = synthetic.make_family(num, length, dim=rt.dim, seed=seed) if family is None else family
        self.coords, self.tensors, self.offsets = synthetic.pack(fam)
        self.lengths = np.diff(self.offsets)
        self.pairs = engine.all_pairs(num)
        self.mine = cdist.partition_pairs(self.pairs, self.lengths, ranks, me) if stride is None else np.arange(len(self.pairs))[::stride]
        self.shard = cdist.shard_size(len(self.pairs), ranks)
        self.gather = rt.use_dist and ranks == rt.world and stride is None
        self.batch = rt.make_batch(self.coords, self.tensors, self.offsets, self.pairs[self.mine])
        self.local = rt.new_local(max(self.shard, len(self.mine)))
        self.gathered_flat = rt.new_gathered(rt.world * self.local.numel()) if self.gather else None

    def step(self, scores_only=False):
        self.rt.run_batch(self.batch, self.local, scores_only)
        if self.gather:
            self.rt.all_gather(self.gathered_flat, self.local)

    def time(self, steps, warmup, collective=True, scores_only=False):
        """seconds per step: `warmup` untimed steps, then `steps` timed ones between fences, max over ranks.
        scores_only: the matrix entries alone (cr_batch_run_scores: what make_pairwise_matrix -> neighbor_joining needs)."""
        rt = self.rt
        for _ in range(warmup):
            self.step(scores_only)
        rt.fence() if collective else rt.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(scores_only)
        rt.fence() if collective else rt.sync()
        el = time.perf_counter() - t0
        return (rt.max_over_ranks(el) if collective else el) / steps

    def gather_ms(self, reps=10):
        """The all-gather's own time per step: events on the launch stream around the collective alone (the share's scores
        already in place), mean of `reps`, max over ranks.  None when this object does not gather."""
        if not self.gather:
            return None
        rt = self.rt
        self.step(True)
        rt.fence()
        ms = [rt.timed_on_stream(lambda: rt.all_gather(self.gathered_flat, self.local)) for _ in range(reps)]
        return rt.max_over_ranks(float(np.mean(ms)) * 1e-3) * 1e3

    def scores(self):
        """This object's score per pair of the WHOLE config (NaN where another rank's pair was not gathered)."""
        rt = self.rt
        from caretta_amd import distributed as cdist
        out = np.full(len(self.pairs), np.nan)
        if self.gather:
            g = rt.to_numpy(self.gathered_flat).reshape(rt.world, -1)
            for r in range(rt.world):
                idx = cdist.partition_pairs(self.pairs, self.lengths, rt.world, r)
                out[idx] = g[r, :len(idx)]
        else:
            out[self.mine] = rt.to_numpy(self.local)[:len(self.mine)]
        return out

    def close(self):
        self.batch.close()


# ----------------------------------------------------------------------------------------------------------------------
# gates
# ----------------------------------------------------------------------------------------------------------------------
def pair_mismatches(gpu_res, gpu_aln, ref, ref_aln, where):
    """Pairs whose GPU results differ from the oracle's: integers exact, floats bit-identical.  `where[k]` = index of the
    oracle's k-th pair in the GPU arrays."""
    mism = 0
    for k, p in enumerate(where):
        ln = int(ref["aln_len"][k])
        ok = int(gpu_res["aln_len"][p]) == ln and np.array_equal(gpu_aln[p, :, :ln], ref_aln[k, :, :ln])
        ok = ok and all(np.array_equal(gpu_res[key][p], ref[key][k]) for key in ("sw", "dtw_score", "rmsd", "tm", "coverage"))
        mism += 0 if ok else 1
    return mism


def oracle_cores(orc):
    return max(1, min(orc.max_threads(), os.cpu_count() or 1))


def pair_gate(orc, coords, tensors, offsets, pairs, gpu_res, gpu_aln, gpu_idx, min_frac=0.01, floor=512):
    """Every output of pipeline H on a sample of at least min_frac of ALL the config's pairs (all of the batch's when they are
    few) drawn from the pairs the GPU batch behind gpu_res / gpu_aln ran (gpu_idx: their indices into `pairs`)."""
    want = max(int(math.ceil(min_frac * len(pairs))), min(len(gpu_idx), floor))
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(len(gpu_idx), size=min(want, len(gpu_idx)), replace=False))
    ref, ref_aln = orc.pairwise_batch(coords, tensors, offsets, pairs[gpu_idx[pick]], want_aln=True, nthreads=oracle_cores(orc))
    return {"checked": int(len(pick)), "of_pairs": int(len(pairs)), "fraction": len(pick) / len(pairs),
            "mismatches": int(pair_mismatches(gpu_res, gpu_aln, ref, ref_aln, pick)),
            "what": "alignment rows and lengths exact; sw, dtw_score, rmsd, tm, coverage bit-identical"}


def nj_tree(matrix):
    """neighbor_joining of max(M) - M (multiple_alignment.py:501) by the library's host implementation."""
    from caretta_amd import neighbor_joining as nj
    tree, _ = nj.neighbor_joining(matrix.max() - matrix)
    return np.asarray(tree)


def config_gate(orc, coords, tensors, offsets, pairs, gpu_scores, gpu_res, gpu_aln, gpu_idx, min_frac=0.01):
    """SURVEY.md 8(d) correctness gate of one BASELINE configuration (all oracle work on all cores):
      * `nj_gate`: the GPU's P x P matrix (gpu_scores, one per pair of `pairs`) against the oracle's matrix of ALL pairs --
        largest difference, neighbor-joining trees identical, bipartition sets identical (neighbor_joining.py:118-129 is
        1-ulp sensitive);
      * `pair_gate`: see pair_gate()."""
    from caretta_amd import engine
    from caretta_amd import neighbor_joining as nj
    num = len(offsets) - 1
    cores = oracle_cores(orc)
    t0 = time.perf_counter()
    full, _ = orc.pairwise_batch(coords, tensors, offsets, pairs, want_aln=False, nthreads=cores)
    t_all = time.perf_counter() - t0
    cpu_matrix = engine.assemble_matrix(pairs, full["sw"], num)
    gpu_matrix = engine.assemble_matrix(pairs, gpu_scores, num)
    t_gpu, _ = nj.neighbor_joining(gpu_matrix.max() - gpu_matrix)                # multiple_alignment.py:501
    t_cpu, _ = orc.neighbor_joining(cpu_matrix.max() - cpu_matrix)
    return {"nj_gate": {"taxa": int(num), "bipartitions_equal": bool(nj.bipartitions(t_gpu, num) == nj.bipartitions(t_cpu, num)),
                        "trees_identical": bool(np.array_equal(t_gpu, t_cpu)),
                        "matrix_max_abs_diff": float(np.abs(gpu_matrix - cpu_matrix).max()),
                        "cpu_matrix": f"C oracle, all {len(pairs)} pairs on {cores} threads in {t_all:.1f} s"},
            "pair_gate": pair_gate(orc, coords, tensors, offsets, pairs, gpu_res, gpu_aln, gpu_idx, min_frac)}


def multi_gpu_gate(pairs, num, gathered_scores, one_gpu_scores, tree_of=nj_tree):
    """The result must not depend on how it is computed (multiple_alignment.py:158-170 has no cross-pair dependency): the
    score vector gathered from all ranks against the one ONE GPU computes for the whole config -- bit for bit (compared as
    64-bit patterns: a NaN can never hide) -- and the neighbor-joining trees of both matrices."""
    from caretta_amd import engine
    a = np.ascontiguousarray(gathered_scores, dtype=np.float64)
    b = np.ascontiguousarray(one_gpu_scores, dtype=np.float64)
    same = a.shape == b.shape and bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)))
    rec = {"pairs": int(len(pairs)), "matrix_equal": same, "nan_scores": int(np.isnan(a).sum()),
           "differing_pairs": int((a.view(np.uint64) != b.view(np.uint64)).sum()) if a.shape == b.shape else int(len(pairs))}
    if rec["nan_scores"] == 0 and not np.isnan(b).any():
        t_a = tree_of(engine.assemble_matrix(pairs, a, num))
        t_b = tree_of(engine.assemble_matrix(pairs, b, num))
        rec["trees_identical"] = bool(np.array_equal(t_a, t_b))
        rec["tree_rows"] = int(t_a.shape[0])
    else:
        rec["trees_identical"] = False
    return rec


def gate_ok(rec) -> bool:
    """Does a record (nested dicts) hold no failed gate?  Keys understood: matrix_equal, trees_identical, bipartitions_equal,
    mismatches, parity_mismatches, streamed_results_equal_fetched, scores_identical."""
    if isinstance(rec, dict):
        for k, v in rec.items():
            if k in ("matrix_equal", "trees_identical", "bipartitions_equal", "streamed_results_equal_fetched", "scores_identical") and v is False:
                return False
            if k in ("mismatches", "parity_mismatches") and isinstance(v, (int, float)) and v != 0:
                return False
            if not gate_ok(v):
                return False
    elif isinstance(rec, (list, tuple)):
        return all(gate_ok(v) for v in rec)
    return True


def failed_gates(rec, path=""):
    """[path, ...] of the failed gates inside a record (for the error message behind a non-zero exit)."""
    out = []
    if isinstance(rec, dict):
        for k, v in rec.items():
            here = f"{path}.{k}" if path else k
            if k in ("matrix_equal", "trees_identical", "bipartitions_equal", "streamed_results_equal_fetched", "scores_identical") and v is False:
                out.append(here)
            elif k in ("mismatches", "parity_mismatches") and isinstance(v, (int, float)) and v != 0:
                out.append(f"{here}={v}")
            else:
                out += failed_gates(v, here)
    elif isinstance(rec, (list, tuple)):
        for i, v in enumerate(rec):
            out += failed_gates(v, f"{path}[{i}]")
    return out


def rank_records(rt):
    """[{rank, device, pci_bus_id, ...}] of every rank (all_gather_object) + the collective library of rank 0."""
    recs = rt.all_gather_object(rt.rank_record())
    recs = sorted(recs, key=lambda r: r["rank"])
    devices = [(r.get("host"), r.get("pci_bus_id") or r.get("device")) for r in recs]
    return {"ranks": recs, "distinct_devices": len(set(devices)), "collective": rt.collective_library()}


def multi_gpu_record(rt, key, steps=5, warmup=2, orc=None, one_gpu=True, family=None, cfg=None):
    """One BASELINE configuration with the pair set FIXED and sharded over the ranks + one all-gather: ms, pairs/s, the
    all-gather's own time; the whole config on ONE GPU (rank 0) for the speed-up and for `multi_gpu_gate`; the oracle's
    `pair_gate` on rank 0's own share.  Every rank calls this; rank 0 gets the record, the others None."""
    n_c, l_c, s_c = cfg or CONFIGS[key]
    sh = Sharded(rt, n_c, l_c, s_c, family=family)
    t_mat = sh.time(steps, warmup, scores_only=True)
    ag_ms = sh.gather_ms()
    t_full = sh.time(steps, warmup)
    gathered = sh.scores() if rt.rank == 0 else None
    share_gate = None
    if rt.rank == 0 and orc is not None:
        r_s, a_s = sh.batch.fetch(want_alignments=True)
        share_gate = pair_gate(orc, sh.coords, sh.tensors, sh.offsets, sh.pairs, r_s, a_s, sh.mine,
                               min_frac=min(0.01, len(sh.mine) / len(sh.pairs)))
        share_gate["what"] = "rank 0's own share against the C oracle: " + share_gate["what"]
    layout = sh.batch.layout()[0]
    pairs, num = sh.pairs, sh.num
    sh.close()
    t_one = t_one_mat = None
    gate = None
    if one_gpu:
        if rt.rank == 0:                                  # the whole config on ONE GPU: the speed-up's base and the gate's other side
            one = Sharded(rt, n_c, l_c, s_c, ranks=1, me=0, family=family)
            t_one_mat = one.time(steps, warmup, collective=False, scores_only=True)
            t_one = one.time(steps, warmup, collective=False)
            gate = multi_gpu_gate(pairs, num, gathered, one.scores())
            one.close()
        rt.fence()
    if rt.rank != 0:
        return None
    npairs = n_c * (n_c - 1) // 2
    rec = {"n_gpus": rt.world, "structures": n_c, "residues": l_c, "pairs": npairs, "ms": t_full * 1e3, "pairs_per_s": npairs / t_full,
           "ms_1gpu": (t_one if t_one is not None else t_full) * 1e3, "speedup_vs_1gpu": (t_one / t_full) if t_one is not None else 1.0,
           "all_gather_ms": ag_ms, "layout_of_share": layout,
           "matrix_only": {"ms": t_mat * 1e3, "ms_1gpu": (t_one_mat if t_one_mat is not None else t_mat) * 1e3,
                           "speedup_vs_1gpu": (t_one_mat / t_mat) if t_one_mat is not None else 1.0}}
    if gate is not None:
        rec["multi_gpu_gate"] = gate
    if share_gate is not None:
        rec["pair_gate_own_share"] = share_gate
    return rec


# ----------------------------------------------------------------------------------------------------------------------
# the CPU baseline (rank 0, N = 1)
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(coords, tensors, offsets, pairs, gpu_res, gpu_aln, gpu_matrix, budget_s=12.0):
    """Time the C oracle (reference-shaped CPU restatement) on a bounded sample of the same pairs and
    use its outputs as the correctness gate for the GPU results; then the whole pair set on all cores for the
    neighbor-joining gate (tree topology of the GPU matrix = tree topology of the CPU matrix)."""
    from caretta_amd import engine
    from caretta_amd import neighbor_joining as nj
    from oracle.pyoracle import Oracle
    orc = Oracle()
    rng = np.random.default_rng(0)
    probe = rng.choice(len(pairs), size=min(8, len(pairs)), replace=False)
    t0 = time.perf_counter()
    orc.pairwise_batch(coords, tensors, offsets, pairs[probe], want_aln=False, nthreads=1)
    per_pair = (time.perf_counter() - t0) / len(probe)
    count = int(min(len(pairs), max(32, budget_s / max(per_pair, 1e-6))))
    sample = np.sort(rng.choice(len(pairs), size=count, replace=False))
    t0 = time.perf_counter()
    ref, ref_aln = orc.pairwise_batch(coords, tensors, offsets, pairs[sample], want_aln=True, nthreads=1)
    t1 = time.perf_counter() - t0
    cores = oracle_cores(orc)
    # correctness gate: integers exact, floats bit-identical (same FP64 operation order on both sides)
    mism = pair_mismatches(gpu_res, gpu_aln, ref, ref_aln, sample)
    out = {
        "value": count / t1, "unit": "pairs/s", "cores": 1, "kind": "port",
        "sample": f"{count} of {len(pairs)} pairs (random, seed 0), C oracle -O2 -ffp-contract=off, reference-shaped "
                  f"(dense f64 DP matrices + int64 backtrack per pair), 1 thread as the reference's pair loop",
        "cpu_model": cpu_model(),
        "parity_mismatches": mism, "parity_checked": int(count),
    }
    # all cores: the WHOLE pair set when it fits ~40 s of CPU time, else a sample (timing only)
    est_all = per_pair * len(pairs) / cores
    whole = est_all <= 40.0
    big = np.arange(len(pairs)) if whole else np.sort(rng.choice(len(pairs), size=min(len(pairs), count * min(cores, 8)), replace=False))
    t0 = time.perf_counter()
    full, _ = orc.pairwise_batch(coords, tensors, offsets, pairs[big], want_aln=False, nthreads=cores)
    tall = time.perf_counter() - t0
    out["all_cores"] = {"value": len(big) / tall, "cores": cores, "pairs": int(len(big))}
    gate = None
    if whole:
        num = len(offsets) - 1
        cpu_matrix = engine.assemble_matrix(pairs, full["sw"], num)
        t_gpu, _ = nj.neighbor_joining(gpu_matrix.max() - gpu_matrix)            # multiple_alignment.py:501
        t_cpu, _ = orc.neighbor_joining(cpu_matrix.max() - cpu_matrix)
        gate = {"taxa": int(num), "bipartitions_equal": bool(nj.bipartitions(t_gpu, num) == nj.bipartitions(t_cpu, num)),
                "trees_identical": bool(np.array_equal(t_gpu, t_cpu)),
                "matrix_max_abs_diff": float(np.abs(gpu_matrix - cpu_matrix).max()),
                "cpu_matrix": f"C oracle, all {len(pairs)} pairs on {cores} threads"}
    return out, gate


# ----------------------------------------------------------------------------------------------------------------------
# the measurement programmes behind the headline (rank-collective unless noted)
# ----------------------------------------------------------------------------------------------------------------------
def incl_transfers_record(rt, head, res, aln, elapsed_per_step, steps):
    """The headline's pair set INCLUDING the PCIe transfers (SURVEY.md 8(d) words the metric this way).  Per step: upload the
    structures (cr_batch_create) and the pair list (cr_batch_set_pairs) from page-locked arrays, run with the download folded
    in (cr_batch_run_stream_i32: the alignment kernel writes every pair's int32 rows and record into page-locked host arrays as
    it finishes the pair), wait.  Nothing is copied after the last kernel.  Then the same work as a PIPELINE, which is how a
    caller with more than one batch would run it: two contexts (two streams) driven by this one host thread -- while batch k
    computes and streams its results out, batch k + 1's structures and pair list are uploaded and its kernels queued on the
    other stream; batch k is waited for after batch k + 1 has been queued.  Every batch still uploads and downloads everything."""
    from caretta_amd import engine
    ctx, params, rank = rt.ctx, rt.params, rt.rank
    coords, tensors, offsets, pairs = head.coords, head.tensors, head.offsets, head.pairs
    my_pairs = pairs[head.mine]
    pin_c, pin_t = engine.pinned_empty(coords.shape, np.float64), engine.pinned_empty(tensors.shape, np.float64)
    pin_c[...], pin_t[...] = coords, tensors           # (what a loader that feeds the GPU would produce)
    pinned = None
    t_parts = np.zeros(2)
    reps = max(3, min(steps, 10))
    streamed_ok = None

    def same_as_fetched(r, a):
        lens = res["aln_len"]
        return bool(r.tobytes() == res.tobytes() and all(np.array_equal(a[p, :, :lens[p]], aln[p, :, :lens[p]]) for p in range(len(lens))))

    for it in range(reps + 2):
        rt.fence()
        t0 = time.perf_counter()
        b2 = engine.PairBatch(ctx, pin_c, pin_t, offsets).set_pairs(my_pairs)
        if pinned is not None:
            b2._pinned_cache = pinned              # result arrays allocated once, as a pipeline would
        t1 = time.perf_counter()
        r2, a2 = b2.run_streamed(params)
        ctx.synchronize()
        t2 = time.perf_counter()
        pinned = b2._pinned_cache
        if it == 0 and rank == 0:                  # what the kernels wrote = what cr_batch_fetch copies
            streamed_ok = same_as_fetched(r2, a2)
        b2.close()
        if it >= 2:
            t_parts += (t1 - t0, t2 - t1)
    t_parts /= reps
    t_serial = rt.max_over_ranks(float(t_parts.sum()))
    ctx_pair = [engine.Context(rt.local_rank), engine.Context(rt.local_rank)]
    caches = [None, None]
    pending = None
    pipelined_ok = True
    rt.fence()
    t0 = None
    for it in range(reps + 3):
        if it == 3:
            ctx_pair[0].synchronize()
            ctx_pair[1].synchronize()
            t0 = time.perf_counter()
        slot = it & 1
        b3 = engine.PairBatch(ctx_pair[slot], pin_c, pin_t, offsets).set_pairs(my_pairs)
        if caches[slot] is not None:
            b3._pinned_cache = caches[slot]
        r3, a3 = b3.run_streamed(params)
        caches[slot] = b3._pinned_cache
        if pending is not None:
            pb, pr, pa, pslot = pending
            ctx_pair[pslot].synchronize()
            if it == 2 and rank == 0:              # a pipelined batch delivers the same bytes
                pipelined_ok = same_as_fetched(pr, pa)
            pb.close()
        pending = (b3, r3, a3, slot)
    pending[0].ctx.synchronize()
    t_pipe = rt.max_over_ranks((time.perf_counter() - t0) / reps)
    pending[0].close()
    for c in ctx_pair:
        c.close()
    if rank != 0:
        return None
    return {"value_incl_transfers": len(pairs) / t_serial, "value_incl_transfers_pipelined": len(pairs) / t_pipe,
            "incl_transfers": {
                "ms_per_step": t_serial * 1e3, "upload_ms": t_parts[0] * 1e3, "run_and_download_ms": t_parts[1] * 1e3,
                "ratio_to_resident": t_serial / elapsed_per_step,
                "pipelined": {"ms_per_step": t_pipe * 1e3, "ratio_to_resident": t_pipe / elapsed_per_step},
                "streamed_results_equal_fetched": bool(streamed_ok and pipelined_ok),
                "downloaded_bytes_per_rank": int(r2.nbytes + a2.nbytes), "uploaded_bytes_per_rank": int(coords.nbytes + tensors.nbytes + my_pairs.nbytes),
                "note": "per batch: cr_batch_create + cr_batch_set_pairs (H2D of structures and pair list from page-locked arrays) and "
                        "cr_batch_run_stream_i32 (the alignment kernel stores all int32 alignment rows + PairResult records into page-locked "
                        "host arrays), a wait after the batch.  pipelined: the same calls as a two-stream pipeline driven by one host thread"}}


def c2_record(rt, orc):
    """BASELINE config 2 (32 x 150): every one of its 496 pairs gated when an oracle is given (N = 1)."""
    n_c, l_c, s_c = CONFIGS["c2"]
    small = Sharded(rt, n_c, l_c, s_c)
    t_small_mat = small.time(20, 3, scores_only=True)
    t_small = small.time(20, 3)
    rec = None
    if rt.rank == 0:
        rec = {"n_gpus": rt.world, "structures": n_c, "residues": l_c, "pairs": len(small.pairs), "ms": t_small * 1e3,
               "pairs_per_s": len(small.pairs) / t_small, "matrix_only_ms": t_small_mat * 1e3}
        if orc is not None and rt.world == 1:
            r_s, a_s = small.batch.fetch(want_alignments=True)
            rec.update(config_gate(orc, small.coords, small.tensors, small.offsets, small.pairs, r_s["sw"], r_s, a_s,
                                   np.arange(len(small.pairs)), min_frac=1.0))
    small.close()
    return rec


def shares_record(rt, key, orc):
    """N = 1: the config on this GPU, then one GPU's share of the 2-, 4- and 8-GPU split (every 2nd / 4th / 8th pair) run here:
    what each of G GPUs would do, before the (latency-bound, ~1 MB) all-gather.  The north star asks throughput at 1, 2, 4 and
    8 GPUs.  With an oracle: the whole config's matrix and neighbor-joining tree against the oracle's, and every output of
    the one-of-8 share's pairs (config 3: ALL of them; else >= 1 % of the config's pairs)."""
    n_c, l_c, s_c = CONFIGS[key]
    sh = Sharded(rt, n_c, l_c, s_c, ranks=1, me=0)
    reps = 10 if key == "c3" else 5
    t_sh_mat = sh.time(reps, 2, collective=False, scores_only=True)
    t_sh = sh.time(reps, 2, collective=False)
    full_scores = sh.scores()
    sh.close()
    npairs = n_c * (n_c - 1) // 2
    rec = {"n_gpus": 1, "structures": n_c, "residues": l_c, "pairs": npairs, "ms": t_sh * 1e3, "pairs_per_s": npairs / t_sh,
           "matrix_only": {"ms": t_sh_mat * 1e3}}
    for g in (2, 4, 8):
        part = Sharded(rt, n_c, l_c, s_c, ranks=1, me=0, stride=g)
        reps_p = (20 if key == "c3" else 10) if g == 8 else (10 if key == "c3" else 5)
        t_part_mat = part.time(reps_p, 3, collective=False, scores_only=True)
        t_part = part.time(reps_p, 3, collective=False)
        rec[f"share_of_{g}"] = {"pairs": int(len(part.mine)), "ms": t_part * 1e3, f"projected_speedup_{g}gpu": t_sh / t_part,
                                "matrix_only_ms": t_part_mat * 1e3, f"matrix_only_projected_speedup_{g}gpu": t_sh_mat / t_part_mat,
                                "layout": part.batch.layout()[0]}
        if orc is not None and g == 8:
            r_p, a_p = part.batch.fetch(want_alignments=True, pinned=True)
            rec.update(config_gate(orc, part.coords, part.tensors, part.offsets, part.pairs, full_scores, r_p, a_p, part.mine,
                                   min_frac=len(part.mine) / len(part.pairs) if key == "c3" else 0.01))
        part.close()
    return rec


def msa_record(rt, num, length, seed, matrix, orc):
    """The consumers behind the matrix: neighbor joining of max(M) - M and the progressive
    alignment of the guide tree, whole tree resident in HBM; with an oracle every tree node is replayed by it."""
    from caretta_amd import multiple_alignment as ma, neighbor_joining as nj, synthetic
    params = rt.params
    fam = synthetic.make_family(num, length, dim=rt.dim, seed=seed)
    prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
    msa = ma.MultipleAlignment(prots)
    sp = dict(flexible=False, gamma_tensor=params.gamma_tensor, gamma_coords=params.gamma_coords, verbose=False)
    t_nj = t_pa = float("inf")
    pa_all = []
    for _ in range(8):
        t0 = time.perf_counter()
        tree, _bl = nj.neighbor_joining(matrix.max() - matrix)
        t1 = time.perf_counter()
        aligned = msa.progressive_align(tree, params.gap_open, params.gap_extend, 1.0, 1.0, sp, dict(flexible=False, verbose=False))
        t2 = time.perf_counter()
        t_nj, t_pa = min(t_nj, t1 - t0), min(t_pa, t2 - t1)
        pa_all.append(t2 - t1)
    rec = {"structures": num, "residues": length, "neighbor_joining_ms": t_nj * 1e3, "progressive_alignment_ms": t_pa * 1e3,
           "progressive_alignment_ms_median": float(np.median(pa_all[1:])) * 1e3, "progressive_alignment_calls": len(pa_all),
           "tree_levels": int(msa.node_table[:, 3].max()), "msa_width": int(len(next(iter(aligned.values())))),
           "note": "MultipleAlignment.progressive_align on the tree of the headline matrix (cr_progressive_align: the whole tree resident in HBM)"}
    if orc is not None:
        # every join replayed by the oracle on the GPU's own child nodes: node coordinates, tensors, weights bit-identical
        tr = np.asarray(tree).astype(np.int64)
        joins = [(int(tr[x, 0]), int(tr[x + 1, 0])) for x in range(0, tr.shape[0] - 1, 2)] + [(int(tr[-1, 0]), int(tr[-1, 1]))]
        sizes, bad = [1] * num, 0
        for k, (n1, n2) in enumerate(joins):
            tot = sizes[n1] + sizes[n2]
            s1, s2 = msa.final_sequences[n1], msa.final_sequences[n2]
            _a1, _a2, xn, tn, wn, _f = orc.progressive_node(s1.coordinates, s1.tensors, msa.final_consensus_weights[n1], s2.coordinates,
                                                           s2.tensors, msa.final_consensus_weights[n2], sizes[n2] / (2 * tot), sizes[n1] / (2 * tot))
            node = msa.final_sequences[num + k]
            bad += int(not (np.array_equal(xn, node.coordinates) and np.array_equal(tn, node.tensors)
                            and np.array_equal(wn, msa.final_consensus_weights[num + k])))
            sizes.append(tot)
        rec["node_gate"] = {"nodes": len(joins), "mismatches": bad, "what": "every tree node replayed by the C oracle on the GPU's "
                            "own children: node coordinates, tensors and consensus weights bit-identical"}
    return rec


def latest_profile(name):
    """The newest committed profiles/rNN/<name> (r06 first), or None."""
    rounds = sorted((p for p in (ROOT / "profiles").glob("r[0-9][0-9]") if (p / name).exists()), reverse=True)
    return (rounds[0] / name) if rounds else None


def shares_summary(extras, world):
    """Compact strong-scaling table, the LAST key of the line (it must survive an 8 KB tail): per BASELINE config, ms of the
    whole config on one GPU and [ms, x] per GPU count -- at N = 1 projected from one GPU's share of the split run alone
    (no all-gather), at N > 1 measured over the N ranks incl. the all-gather, with the gate verdicts."""
    out = {"what": ("ms one GPU; per G: [ms of one GPU's share run alone, projected x]" if world == 1 else
                    f"ms one GPU; measured on {world} ranks incl. all-gather: [ms, x, all_gather_ms, matrix_equal, trees_identical, share_mismatches]")}
    for key in ("c3", "c4", "c5"):
        rec = extras.get(f"{key}_sharded")
        if not rec:
            continue
        row = {"ms_1gpu": round(rec.get("ms_1gpu", rec["ms"]), 4)}
        if "share_of_8" in rec:
            for g in (2, 4, 8):
                s = rec[f"share_of_{g}"]
                row[str(g)] = [round(s["ms"], 4), round(s[f"projected_speedup_{g}gpu"], 3)]
        if rec.get("n_gpus", 1) > 1 or "multi_gpu_gate" in rec:
            g = rec.get("multi_gpu_gate", {})
            row[f"measured_{world}"] = [round(rec.get("forced_dist_ms", rec["ms"]), 4), round(rec.get("speedup_vs_1gpu", 1.0), 3),
                                               None if rec.get("all_gather_ms") is None else round(rec["all_gather_ms"], 4),
                                               g.get("matrix_equal"), g.get("trees_identical"),
                                               rec.get("pair_gate_own_share", {}).get("mismatches")]
        out[key] = row
    return out
