#!/usr/bin/env python3
"""Benchmark of the all-vs-all pairwise structural alignment path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: starts its own N rank processes, one per GPU; under
                                                          torch.distributed.run it is one of the ranks already)

A "step" is one pass of the pairwise pipeline (seed fill -> seed traceback+Kabsch -> alignment fill
-> alignment traceback+metrics, plus the all-gather of the score vectors when N>1) over the whole
pair set of a synthetic family whose structures are already resident in HBM.  Prints ONE JSON line.

Workloads (BASELINE.json configs; d=10, seeds 20240+k):
  headline : P = round(128*sqrt(N)) structures x 300 residues, all pairs sharded over N GPUs
             (N=1 is BASELINE config 3: 128 x 300 = 8128 pairs; per-GPU work is fixed -> weak scaling)
  c2       : 32 x 150      c4 : 512 x 300      c5 : 64 x 1200

Besides `value` (device-resident rate, SURVEY.md 8(d) and the driver's contract; the timed region of K steps is REPEATED
`--repeats` times and `ms_per_step` / `value` are the MEDIAN run's, min / median / max in `repeats`) the line carries
  value_strong         : (N > 1) BASELINE config 3's own 128 x 300 with the pair set FIXED and sharded over the N ranks, pairs/s --
                         `value` itself is the WEAK headline (P grows with sqrt(N)); `config.workload` says so;
  value_incl_transfers : the same pair set INCLUDING the upload of the structures and the download of every result
                         (alignment rows, transforms, metrics), one batch with a wait behind it -- the metric as SURVEY.md 8(d)
                         words it; value_incl_transfers_pipelined: the steady state of a two-stream pipeline of such batches;
  ragged, mixed        : what every real input is -- 160 structures of 80 .. 520 residues, all 12 720 pairs (ms, Mcells/s beside the
                         headline's, >= 1 % of the pairs gated against the oracle), and a 150-residue family with two 600-residue
                         members as one list against the sum of its homogeneous parts (tools/ragged_time.py);
  explicit_batch       : the reference's functions on EXPLICIT score matrices over the 8 128 x 300 x 300 list, with and without
                         their tracebacks: ms, GB/s of SURVEY 8(d)'s explicit-mode bytes (tools/explicit_batch_rate.py);
  c3_sharded, c4_sharded, c5_sharded : BASELINE configs 3, 4 and 5 timed in the same run with the pair set FIXED and sharded
                         over the N ranks + one all-gather, with the speed-up against ONE GPU running the whole config
                         (measured on rank 0) -- c3_sharded at N > 1 is the strong-scaling record of the headline's own 128 x 300;
                         at N=1 also `share_of_2`, `share_of_4`, `share_of_8`: one GPU's share of the 2-, 4-, 8-GPU split run on this GPU;
  matrix_only          : the P x P matrix entries alone (cr_batch_run_scores), what make_pairwise_matrix -> NJ consumes;
  nj_gate              : neighbor-joining bipartitions of the GPU matrix = those of the all-core oracle matrix (N=1);
  msa                  : the consumers behind the matrix at N=1: neighbor joining and the progressive alignment of the guide
                         tree (ms), every tree node replayed by the oracle (node_gate);
  roofline, cpu_baseline : as the contract asks (roofline.frac from SURVEY 8(d)'s algorithmic bytes).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64, spec; counts an FMA as 2
CONFIGS = {"c2": (32, 150, 20241), "c3": (128, 300, 20242), "c4": (512, 300, 20243), "c5": (64, 1200, 20244)}


def workload(name: str, n_gpus: int):
    if name == "headline":
        return int(round(128 * math.sqrt(n_gpus))), 300, 20242
    return CONFIGS[name]


def stage_bytes(lengths, pairs, d):
    """Algorithmic HBM bytes per launch of the two fill kernels, SURVEY.md 8(d) / DESIGN.md section 5:
    k_seed reads the two structures' tensors and writes 2 bits per cell (70 500 B per 300 x 300 pair, d = 10); k_align reads the
    coordinates and writes 4 bits per cell, the two alignment rows and the pair's record (69 136 B).  `*_readback` adds the traceback's re-read of the decision words and the small
    per-pair records (the round-1 figure)."""
    n = lengths[pairs[:, 0]].astype(np.float64)
    m = lengths[pairs[:, 1]].astype(np.float64)
    seed = 8.0 * d * (n + m) + n * m / 4
    align = 24.0 * (n + m) + n * m / 2 + 16.0 * (n + m) + 136.0       # (coordinates in; 4-bit decisions, two int64 rows, the 136-byte record out)
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
    cores = max(1, min(orc.max_threads(), os.cpu_count() or 1))
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


def config_gate(orc, coords, tensors, offsets, pairs, gpu_scores, gpu_res, gpu_aln, gpu_idx, min_frac=0.01):
    """SURVEY.md 8(d) correctness gate of one BASELINE configuration (all oracle work on all cores):
      * `nj_gate`: the GPU's P x P matrix (gpu_scores, one per pair of `pairs`) against the oracle's matrix of ALL pairs --
        largest difference, neighbor-joining trees identical, bipartition sets identical (neighbor_joining.py:118-129 is
        1-ulp sensitive);
      * `pair_gate`: every output of pipeline H on a sample of at least min_frac of all pairs (all of them when they are
        few) drawn from the pairs the GPU batch behind gpu_res / gpu_aln ran (gpu_idx: their indices into `pairs`)."""
    from caretta_amd import engine
    from caretta_amd import neighbor_joining as nj
    num = len(offsets) - 1
    cores = max(1, min(orc.max_threads(), os.cpu_count() or 1))
    t0 = time.perf_counter()
    full, _ = orc.pairwise_batch(coords, tensors, offsets, pairs, want_aln=False, nthreads=cores)
    t_all = time.perf_counter() - t0
    cpu_matrix = engine.assemble_matrix(pairs, full["sw"], num)
    gpu_matrix = engine.assemble_matrix(pairs, gpu_scores, num)
    t_gpu, _ = nj.neighbor_joining(gpu_matrix.max() - gpu_matrix)                # multiple_alignment.py:501
    t_cpu, _ = orc.neighbor_joining(cpu_matrix.max() - cpu_matrix)
    gate = {"nj_gate": {"taxa": int(num), "bipartitions_equal": bool(nj.bipartitions(t_gpu, num) == nj.bipartitions(t_cpu, num)),
                        "trees_identical": bool(np.array_equal(t_gpu, t_cpu)),
                        "matrix_max_abs_diff": float(np.abs(gpu_matrix - cpu_matrix).max()),
                        "cpu_matrix": f"C oracle, all {len(pairs)} pairs on {cores} threads in {t_all:.1f} s"}}
    want = max(int(math.ceil(min_frac * len(pairs))), min(len(gpu_idx), 512))
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(len(gpu_idx), size=min(want, len(gpu_idx)), replace=False))
    ref, ref_aln = orc.pairwise_batch(coords, tensors, offsets, pairs[gpu_idx[pick]], want_aln=True, nthreads=cores)
    gate["pair_gate"] = {"checked": int(len(pick)), "of_pairs": int(len(pairs)), "fraction": len(pick) / len(pairs),
                         "mismatches": int(pair_mismatches(gpu_res, gpu_aln, ref, ref_aln, pick)),
                         "what": "alignment rows and lengths exact; sw, dtw_score, rmsd, tm, coverage bit-identical"}
    return gate


def spawn_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start one rank process per GPU (children of this process, which
    itself makes no GPU call and imports neither torch nor the HIP library), hand rank 0's JSON line on, and return
    non-zero if any rank fails.  Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None                                   # set when a rank has failed: the others may be stuck in a collective
    while any(p.poll() is None for p in procs):
        if deadline is None and any(p.poll() not in (None, 0) for p in procs):
            deadline = time.monotonic() + 15.0
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                          # exactly the processes started above
        time.sleep(0.05)
    reader.join(timeout=5)
    codes = [p.returncode for p in procs]
    sys.stdout.write(b"".join(chunks).decode("utf-8", "replace"))
    sys.stdout.flush()
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=["headline", "c2", "c3", "c4", "c5"])
    ap.add_argument("--repeats", type=int, default=10, help="how often the timed region of --steps steps is repeated (the line reports the median run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_incl_transfers and the c4/c5 sharded timings")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process starts the N ranks itself (it never touches a GPU)
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: caretta_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CARETTA_FORCE_DIST=1 runs the RCCL code path (process group, all-gather, barrier, max-reduce) even with a single
    # rank, so that it can be exercised on a 1-GPU box under torch.distributed.run
    use_dist = world > 1 or (os.environ.get("CARETTA_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if use_dist:
        dist.barrier()

    from caretta_amd import distributed as cdist
    from caretta_amd import engine, synthetic

    dim = 10
    # the kernels run on torch's current stream (the legacy default stream, handle 0, unless the caller changed it):
    # the all-gather that follows cr_batch_run is ordered behind the kernels by the stream itself
    stream = torch.cuda.current_stream(dev)
    ctx = engine.Context(local_rank, stream=stream.cuda_stream)
    params = engine.make_params()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(seconds: float) -> float:
        if not use_dist:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    class Sharded:
        """One config's pair set sharded over the ranks: batch on this rank's share + the all-gather buffers."""

        def __init__(self, num, length, seed, ranks=world, me=rank, stride=None):
            self.num, self.length, self.seed = num, length, seed
            fam = synthetic.make_family(num, length, dim=dim, seed=seed)
            self.coords, self.tensors, self.offsets = synthetic.pack(fam)
            self.lengths = np.diff(self.offsets)
            self.pairs = engine.all_pairs(num)
            self.mine = cdist.partition_pairs(self.pairs, self.lengths, ranks, me) if stride is None else np.arange(len(self.pairs))[::stride]
            self.shard = cdist.shard_size(len(self.pairs), ranks)
            self.gather = use_dist and ranks == world and stride is None
            self.batch = engine.PairBatch(ctx, self.coords, self.tensors, self.offsets).set_pairs(self.pairs[self.mine])
            self.local = torch.full((max(self.shard, len(self.mine)),), float("nan"), dtype=torch.float64, device=dev)
            self.gathered_flat = torch.empty(world * self.local.numel(), dtype=torch.float64, device=dev) if self.gather else None

        def step(self, scores_only=False):
            self.batch.run(params, sw_out_device_ptr=self.local.data_ptr(), scores_only=scores_only)
            if self.gather:
                dist.all_gather_into_tensor(self.gathered_flat, self.local)

        def time(self, steps, warmup, collective=True, scores_only=False):
            """seconds per step: `warmup` untimed steps, then `steps` timed ones between fences, max over ranks.
            scores_only: the matrix entries alone (cr_batch_run_scores: what make_pairwise_matrix -> neighbor_joining needs)."""
            for _ in range(warmup):
                self.step(scores_only)
            fence() if collective else torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(scores_only)
            fence() if collective else torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            return (max_over_ranks(el) if collective else el) / steps

        def close(self):
            self.batch.close()

    # ------------------------------------------------------------------ the headline, timed as the contract says
    num, length, seed = workload(args.workload, args.gpus)
    head = Sharded(num, length, seed)
    for _ in range(args.warmup):
        head.step()
    fence()
    repeats = max(1, args.repeats)
    ctx.set_profiling(min(args.steps * repeats, 4096))
    run_s = []
    for _ in range(repeats):
        # one timed region as the contract words it: EXACTLY `steps` steps between barrier + synchronize, max over ranks
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            head.step()
        fence()
        run_s.append(max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(run_s))                  # the median run's time is the line's ms_per_step / value
    stage_ms, runs = head.batch.stage_ms()
    ctx.set_profiling(0)
    gathered = head.gathered_flat.view(world, -1) if head.gather else head.local.unsqueeze(0)
    res, aln = head.batch.fetch(want_alignments=(rank == 0))
    matrix = cdist.scatter_to_matrix(gathered.cpu().numpy(), head.pairs, head.lengths, num)
    pairs, lengths, mine = head.pairs, head.lengths, head.mine
    coords, tensors, offsets = head.coords, head.tensors, head.offsets

    extras = {}
    if not args.no_extras:
        # ---------------------------------------------------------- the P x P matrix alone (no pairwise alignments)
        t_mat = head.time(max(3, min(args.steps, 10)), 2, scores_only=True)
        if rank == 0:
            extras["matrix_only"] = {"ms_per_step": t_mat * 1e3, "pairs_per_s": len(pairs) / t_mat,
                                     "note": "cr_batch_run_scores: seed kernel + smith_waterman_score of the coordinate score matrix per pair "
                                             "(multiple_alignment.py:158-170), no dtw_align / traceback / metrics"}
        # ---------------------------------------------------------- the same pair set including the PCIe transfers
        # (SURVEY.md 8(d) words the metric this way).  Per step: upload the structures (cr_batch_create) and the pair list
        # (cr_batch_set_pairs) from page-locked arrays, run with the download folded in (cr_batch_run_stream_i32: the
        # alignment kernel writes every pair's int32 rows and record into page-locked host arrays as it finishes the
        # pair), wait.  Nothing is copied after the last kernel.
        my_pairs = pairs[mine]
        pin_c, pin_t = engine.pinned_empty(coords.shape, np.float64), engine.pinned_empty(tensors.shape, np.float64)
        pin_c[...], pin_t[...] = coords, tensors           # (what a loader that feeds the GPU would produce)
        pinned = None
        t_parts = np.zeros(2)
        reps = max(3, min(args.steps, 10))
        streamed_ok = None
        for it in range(reps + 2):
            fence()
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
                lens = res["aln_len"]
                streamed_ok = bool(r2.tobytes() == res.tobytes()
                                   and all(np.array_equal(a2[p, :, :lens[p]], aln[p, :, :lens[p]]) for p in range(len(lens))))
            b2.close()
            if it >= 2:
                t_parts += (t1 - t0, t2 - t1)
        t_parts /= reps
        t_serial = max_over_ranks(float(t_parts.sum()))
        # The same work as a PIPELINE, which is how a caller with more than one batch would run it: two contexts (two streams)
        # driven by this one host thread -- while batch k computes and streams its results out, batch k + 1's structures and
        # pair list are uploaded and its kernels queued on the other stream; batch k is waited for (its results are then
        # complete in ITS page-locked arrays) after batch k + 1 has been queued.  Every batch still uploads everything and
        # downloads everything; the time per batch is the steady state over `reps` batches.
        ctx_pair = [engine.Context(local_rank), engine.Context(local_rank)]
        caches = [None, None]
        pending = None
        pipelined_ok = True
        fence()
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
                    lens = res["aln_len"]
                    pipelined_ok = bool(pr.tobytes() == res.tobytes()
                                        and all(np.array_equal(pa[p, :, :lens[p]], aln[p, :, :lens[p]]) for p in range(len(lens))))
                pb.close()
            pending = (b3, r3, a3, slot)
        pending[0].ctx.synchronize()
        t_pipe = max_over_ranks((time.perf_counter() - t0) / reps)
        pending[0].close()
        for c in ctx_pair:
            c.close()
        if rank == 0:
            # (value_incl_transfers is ONE batch with a wait behind it, as in rounds 1-3 and as SURVEY 8(d) words the metric;
            # round 4's line carried the pipelined figure under this key)
            extras["value_incl_transfers"] = len(pairs) / t_serial
            extras["value_incl_transfers_pipelined"] = len(pairs) / t_pipe
            extras["incl_transfers"] = {
                "ms_per_step": t_serial * 1e3, "upload_ms": t_parts[0] * 1e3, "run_and_download_ms": t_parts[1] * 1e3,
                "ratio_to_resident": t_serial / (elapsed / args.steps),
                "pipelined": {"ms_per_step": t_pipe * 1e3, "ratio_to_resident": t_pipe / (elapsed / args.steps)},
                "streamed_results_equal_fetched": bool(streamed_ok and pipelined_ok),
                "downloaded_bytes_per_rank": int(r2.nbytes + a2.nbytes), "uploaded_bytes_per_rank": int(coords.nbytes + tensors.nbytes + my_pairs.nbytes),
                "note": "per batch: cr_batch_create + cr_batch_set_pairs (H2D of structures and pair list from page-locked arrays) and "
                        "cr_batch_run_stream_i32 (the alignment kernel stores all int32 alignment rows + PairResult records into page-locked "
                        "host arrays), a wait after the batch.  pipelined: the same calls as a two-stream pipeline driven by one host thread -- "
                        "batch k + 1 is uploaded and queued while batch k computes; every batch uploads and downloads everything"}
        # ---------------------------------------------------------- BASELINE configs 4 and 5, sharded over the ranks
        gated = world == 1 and not args.no_cpu_baseline
        orc = None
        if gated:
            from oracle.pyoracle import Oracle
            orc = Oracle()
        if args.workload != "c2":
            # ------------------------------------------------------ BASELINE config 2: every one of the 496 pairs gated
            n_c, l_c, s_c = CONFIGS["c2"]
            small = Sharded(n_c, l_c, s_c)
            t_small_mat = small.time(20, 3, scores_only=True)
            t_small = small.time(20, 3)
            if rank == 0:
                rec = {"n_gpus": world, "structures": n_c, "residues": l_c, "pairs": len(small.pairs), "ms": t_small * 1e3,
                       "pairs_per_s": len(small.pairs) / t_small, "matrix_only_ms": t_small_mat * 1e3}
                if gated:
                    r_s, a_s = small.batch.fetch(want_alignments=True)
                    rec.update(config_gate(orc, small.coords, small.tensors, small.offsets, small.pairs, r_s["sw"], r_s, a_s,
                                           np.arange(len(small.pairs)), min_frac=1.0))
                extras["c2"] = rec
            small.close()
        # (c3 = the headline's own 128 x 300 with the pair set FIXED: at N > 1 its record is the strong-scaling figure of the
        # north star -- ">= 6x further scaling at 8 GPUs" -- next to the weak `value`; at N = 1 its share_of_8 is what one of
        # eight GPUs would run: 1 016 pairs, the mid-size layout of cr_duo.h)
        for key in ("c3", "c4", "c5"):
            n_c, l_c, s_c = CONFIGS[key]
            sh = Sharded(n_c, l_c, s_c)
            t_sh_mat = sh.time(10 if key == "c3" else 5, 2, scores_only=True)
            t_sh = sh.time(10 if key == "c3" else 5, 2)
            full_scores = sh.local[:len(sh.pairs)].cpu().numpy() if world == 1 else None
            sh.close()
            t_one = t_one_mat = None
            if world > 1:
                if rank == 0:                           # the whole config on ONE GPU, for the speed-up
                    one = Sharded(n_c, l_c, s_c, ranks=1, me=0)
                    t_one = one.time(5, 2, collective=False)
                    t_one_mat = one.time(5, 2, collective=False, scores_only=True)
                    one.close()
                fence()
            rec = {"n_gpus": world, "structures": n_c, "residues": l_c, "pairs": n_c * (n_c - 1) // 2,
                   "ms": t_sh * 1e3, "pairs_per_s": n_c * (n_c - 1) / 2 / t_sh,
                   "ms_1gpu": (t_one if t_one is not None else t_sh) * 1e3,
                   "speedup_vs_1gpu": (t_one / t_sh) if t_one is not None else 1.0,
                   "matrix_only": {"ms": t_sh_mat * 1e3, "ms_1gpu": (t_one_mat if t_one_mat is not None else t_sh_mat) * 1e3,
                                   "speedup_vs_1gpu": (t_one_mat / t_sh_mat) if t_one_mat is not None else 1.0}}
            if world == 1:
                # one GPU's share of the 2-, 4- and 8-GPU split (every 2nd / 4th / 8th pair), run here: what G GPUs would each do,
                # before the (latency-bound, ~1 MB) all-gather.  The north star asks throughput at 1, 2, 4 and 8 GPUs.
                for g in (2, 4, 8):
                    part = Sharded(n_c, l_c, s_c, ranks=1, me=0, stride=g)
                    reps_p = (20 if key == "c3" else 10) if g == 8 else (10 if key == "c3" else 5)
                    t_part_mat = part.time(reps_p, 3, collective=False, scores_only=True)
                    t_part = part.time(reps_p, 3, collective=False)
                    rec[f"share_of_{g}"] = {"pairs": int(len(part.mine)), "ms": t_part * 1e3, f"projected_speedup_{g}gpu": t_sh / t_part,
                                            "matrix_only_ms": t_part_mat * 1e3, f"matrix_only_projected_speedup_{g}gpu": t_sh_mat / t_part_mat,
                                            "layout": part.batch.layout()[0]}
                    if gated and g == 8:
                        # the whole config's matrix (the one-GPU run above) against the oracle's, and every output of the
                        # share's pairs (>= 1 % of the config's pairs) against the oracle's
                        r_p, a_p = part.batch.fetch(want_alignments=True, pinned=True)
                        # (c3: EVERY pair of the share against the oracle, not a sample)
                        rec.update(config_gate(orc, part.coords, part.tensors, part.offsets, part.pairs, full_scores, r_p, a_p, part.mine,
                                               min_frac=len(part.mine) / len(part.pairs) if key == "c3" else 0.01))
                    part.close()
            extras[f"{key}_sharded"] = rec

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "headline":
        # ---------------------------------------------------------- the consumers behind the matrix (SURVEY 8f, rows f-1,
        # a19): neighbor joining of max(M) - M and the progressive alignment of the guide tree, whole tree resident in HBM
        from caretta_amd import multiple_alignment as ma, neighbor_joining as nj
        fam = synthetic.make_family(num, length, dim=dim, seed=seed)
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
               "note": "MultipleAlignment.progressive_align on the tree of the headline matrix (cr_progressive_align: per tree "
                       "level the RBF scores of all nodes by their own launches, SW / affine-DTW sweeps on them, node merge)"}
        if gated:
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
        extras["msa"] = rec

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "headline":
        # ---------------------------------------------------------- what the reference actually runs: ragged families, and its
        # functions on explicit score matrices WITH their tracebacks (dynamic_time_warping.py:148-184, :226-278)
        sys.path.insert(0, str(ROOT / "tools"))
        try:
            import ragged_time
            rec = ragged_time.ragged_record(ctx, params, orc)
            rec["headline_mcells_per_s"] = float((lengths[pairs[:, 0]] * lengths[pairs[:, 1]]).sum()) / (elapsed / args.steps) / 1e6
            rec["mcells_per_s_over_headline"] = rec["mcells_per_s"] / rec["headline_mcells_per_s"]
            extras["ragged"] = rec
            extras["mixed"] = ragged_time.mixed_record(ctx, params)
        except Exception as exc:                          # noqa: BLE001 -- the headline line must survive
            extras["ragged"] = {"error": repr(exc)}
        try:
            import explicit_batch_rate
            extras["explicit_batch"] = explicit_batch_rate.explicit_record(8128, 300)
            tfile = ROOT / "profiles" / "r05" / "explicit_batch_pmc.json"
            if tfile.exists():
                extras["explicit_batch"]["traffic_over_algorithmic"] = json.loads(tfile.read_text()).get("traffic_over_algorithmic")
                extras["explicit_batch"]["traffic_source"] = "profiles/r05/explicit_batch_pmc.json"
        except Exception as exc:                          # noqa: BLE001
            extras["explicit_batch"] = {"error": repr(exc)}

    if rank == 0 and world == 1 and not args.no_extras and engine.device_count() > 1:
        # several GPUs visible to this ONE process: the single-process multi-GPU path behind make_pairwise_matrix
        # (cr_multi_*) on BASELINE config 4, in a child process with a time limit (tools/multi_gpu_check.py)
        import subprocess
        try:
            # (the tool is a watchdog: a fresh child per measurement under its own time limit, non-zero exit on timeout or difference)
            p = subprocess.run([sys.executable, str(ROOT / "tools" / "multi_gpu_check.py"), "--timeout", "240"], capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            rec_m = json.loads(line[-1]) if line else {}
            if p.returncode != 0:
                rec_m.setdefault("error", f"exit code {p.returncode}")
                rec_m["stderr_tail"] = p.stderr[-500:]
            extras["single_process_multi_gpu"] = rec_m
        except Exception as exc:                          # noqa: BLE001 -- the headline line must survive
            extras["single_process_multi_gpu"] = {"error": repr(exc)}
    if rank == 0:
        total_pairs = len(pairs)
        ms_per_step = elapsed / args.steps * 1e3
        sb = stage_bytes(lengths, pairs[mine], dim)
        dom = 1 if stage_ms[1] >= stage_ms[0] else 0
        dom_name = "k_align" if dom == 1 else "k_seed"
        dom_bytes = sb[dom_name]
        achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
        achieved_rb = sb[dom_name + "_readback"] / (stage_ms[dom] * 1e-3) / 1e9
        cells_rank = float((lengths[pairs[mine][:, 0]] * lengths[pairs[mine][:, 1]]).sum())
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text())
                traffic = rec.get(f"{args.workload}:{args.gpus}:{dom_name}", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "pairwise alignments/sec", "value": total_pairs / (elapsed / args.steps), "unit": "pairs/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if args.workload == "headline" else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": {"runs": repeats, "steps_per_run": args.steps, "ms_per_step_min": min(run_s) / args.steps * 1e3,
                        "ms_per_step_median": elapsed / args.steps * 1e3, "ms_per_step_max": max(run_s) / args.steps * 1e3,
                        "note": "the timed region (K steps between barrier + synchronize) repeated; ms_per_step / value are the median run's"},
            "config": {"workload": f"{args.workload}: {num} structures x {length} residues, d={dim}, all {total_pairs} "
                                   f"pairs i<j sharded over {args.gpus} GPU(s), pipeline H (tensor-RBF SW seed -> Kabsch -> "
                                   f"coord-RBF SW score + affine DTW(1.0,0.01) -> Kabsch/RMSD/TM)"
                                   + (f"; `value` is the WEAK-scaling headline (P = round(128 sqrt(N)) = {num}: per-GPU work fixed); the "
                                      f"STRONG-scaling figure of BASELINE config 3 (128 x 300 fixed) is `value_strong`"
                                      if args.workload == "headline" and args.gpus > 1 else ""),
                       "structures": num, "residues": length, "tensor_width": dim, "pairs": total_pairs,
                       "pairs_per_gpu": int(len(mine)), "seed": seed},
            "dtw_mcells_per_s": cells_rank * world / (stage_ms[1] * 1e-3) / 1e6,
            "stage_ms": {"k_seed": stage_ms[0], "k_align": stage_ms[1], "runs_averaged": runs},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": stage_ms[dom],
                         "frac_with_readback": achieved_rb / HBM_PEAK_GBS,
                         "algorithmic_bytes_with_readback": sb[dom_name + "_readback"],
                         "note": "SURVEY 8(d) bytes: features in + packed decisions out.  The fused RBF+DP kernels are bound by "
                                 "FP64-rate VALU issue, not HBM (DESIGN.md section 5; valu_f64 below); the HBM-bound kernel of the "
                                 "path is the batched explicit-matrix row sweep (profiles/r04/explicit_batch_rate.txt)"},
            "valu_f64": {"est_flop_per_cell": {"seed_fill": 59, "align_fill": 49},
                         "achieved_tflops": (59 + 49) * cells_rank / ((stage_ms[0] + stage_ms[1]) * 1e-3) / 1e12,
                         "peak_tflops": FP64_VALU_PEAK_TFLOPS},
        }
        # the binding resource, from the committed PMC summary of this workload (tools/pmc_profile.sh): share of all
        # cycles in which a SIMD issues a VALU instruction = SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs)
        try:
            if args.workload == "headline" and args.gpus == 1:
                for rnd in ("r04", "r03", "r02", "r01"):
                    f = ROOT / "profiles" / rnd / "pmc_summary.json"
                    if f.exists():
                        pmc = json.load(open(f))
                        out["valu_f64"]["issue_frac_pmc"] = {
                            k.split("<")[0]: round(v["SQ_INSTS_VALU"] / 1024 * 4 / (v["GRBM_GUI_ACTIVE"] / 8), 4)
                            for k, v in pmc.items()
                            if k.split("<")[0] in ("k_seed", "k_align") and "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v}
                        out["valu_f64"]["issue_frac_pmc"]["source"] = f"profiles/{rnd}/pmc_summary.json"
                        break
        except (OSError, ValueError, KeyError):
            pass
        out.update(extras)
        if args.gpus > 1 and "c3_sharded" in extras:
            out["value_strong"] = extras["c3_sharded"]["pairs_per_s"]
            out["value_strong_note"] = "BASELINE config 3 (128 x 300, 8 128 pairs) FIXED and sharded over the ranks + one all-gather: pairs/s; speed-up vs one GPU in c3_sharded.speedup_vs_1gpu"
        out["build"] = ge.build_provenance()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["nj_gate"] = cpu_baseline(coords, tensors, offsets, pairs, res, aln, matrix)
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        out["matrix_checksum"] = float(matrix.sum())
        print(json.dumps(out))
    head.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
