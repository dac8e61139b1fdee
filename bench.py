#!/usr/bin/env python3
"""Benchmark of the all-vs-all pairwise structural alignment path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one pass of the pairwise pipeline (seed fill -> seed traceback+Kabsch -> alignment fill
-> alignment traceback+metrics, plus the all-gather of the score vectors when N>1) over the whole
pair set of a synthetic family whose structures are already resident in HBM.  Prints ONE JSON line.

Workloads (BASELINE.json configs; d=10, seeds 20240+k):
  headline : P = round(128*sqrt(N)) structures x 300 residues, all pairs sharded over N GPUs
             (N=1 is BASELINE config 3: 128 x 300 = 8128 pairs; per-GPU work is fixed -> weak scaling)
  c2       : 32 x 150      c4 : 512 x 300      c5 : 64 x 1200
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


def workload(name: str, n_gpus: int):
    if name == "headline":
        return int(round(128 * math.sqrt(n_gpus))), 300, 20242
    return {"c2": (32, 150, 20241), "c3": (128, 300, 20242), "c4": (512, 300, 20243), "c5": (64, 1200, 20244)}[name]


def stage_bytes(lengths, pairs, d):
    """Algorithmic HBM bytes per launch of the two fill kernels."""
    n = lengths[pairs[:, 0]].astype(np.float64)
    m = lengths[pairs[:, 1]].astype(np.float64)
    # k_seed: tensors in, 2-bit decisions out and read back by the traceback, aligned coords, transform out
    seed = 8.0 * d * (n + m) + 2 * (n * m / 4) + 24.0 * (n + m) + 144
    # k_align: coords + transform in, 4-bit decisions out and back, alignment rows + results out
    align = 24.0 * (n + m) + 144 + 2 * (n * m / 2) + 8.0 * (n + m) + 160
    return float(seed.sum()), float(align.sum())


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(coords, tensors, offsets, pairs, gpu_res, gpu_aln, budget_s=12.0):
    """Time the C oracle (reference-shaped CPU restatement) on a bounded sample of the same pairs and
    use its outputs as the correctness gate for the GPU results."""
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
    big = np.sort(rng.choice(len(pairs), size=min(len(pairs), count * min(cores, 8)), replace=False))
    t0 = time.perf_counter()
    orc.pairwise_batch(coords, tensors, offsets, pairs[big], want_aln=False, nthreads=cores)
    tall = time.perf_counter() - t0
    # correctness gate: integers exact, floats bit-identical (same FP64 operation order on both sides)
    mism = 0
    for k, p in enumerate(sample):
        ln = int(ref["aln_len"][k])
        ok = int(gpu_res["aln_len"][p]) == ln and np.array_equal(gpu_aln[p, :, :ln], ref_aln[k, :, :ln])
        ok = ok and all(np.array_equal(gpu_res[key][p], ref[key][k]) for key in ("sw", "dtw_score", "rmsd", "tm", "coverage"))
        mism += 0 if ok else 1
    return {
        "value": count / t1, "unit": "pairs/s", "cores": 1, "kind": "port",
        "sample": f"{count} of {len(pairs)} pairs (random, seed 0), C oracle -O2 -ffp-contract=off, reference-shaped "
                  f"(dense f64 DP matrices + int64 backtrack per pair), 1 thread as the reference's pair loop",
        "all_cores": {"value": len(big) / tall, "cores": cores, "pairs": int(len(big))},
        "cpu_model": cpu_model(),
        "parity_mismatches": mism, "parity_checked": int(count),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=["headline", "c2", "c3", "c4", "c5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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

    num, length, seed = workload(args.workload, args.gpus)
    dim = 10
    fam = synthetic.make_family(num, length, dim=dim, seed=seed)
    coords, tensors, offsets = synthetic.pack(fam)
    lengths = np.diff(offsets)
    pairs = engine.all_pairs(num)
    mine = cdist.partition_pairs(pairs, lengths, world, rank)
    shard = cdist.shard_size(len(pairs), world)

    stream = torch.cuda.current_stream(dev)
    ctx = engine.Context(local_rank, stream=stream.cuda_stream)
    batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs[mine])
    params = engine.make_params()
    local = torch.full((shard,), float("nan"), dtype=torch.float64, device=dev)
    gathered_flat = torch.empty(world * shard, dtype=torch.float64, device=dev)
    gathered = gathered_flat.view(world, shard)

    def step():
        batch.run(params, sw_out_device_ptr=local.data_ptr())
        if use_dist:
            dist.all_gather_into_tensor(gathered_flat, local)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    ctx.set_profiling(min(args.steps, 4096))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms, runs = batch.stage_ms()
    if not use_dist:
        gathered.copy_(local.unsqueeze(0))

    res, aln = batch.fetch(want_alignments=(rank == 0))
    matrix = cdist.scatter_to_matrix(gathered.cpu().numpy(), pairs, lengths, num)

    if rank == 0:
        total_pairs = len(pairs)
        ms_per_step = elapsed / args.steps * 1e3
        seed_b, align_b = stage_bytes(lengths, pairs[mine], dim)
        dom = 1 if stage_ms[1] >= stage_ms[0] else 0
        dom_name = "k_align" if dom == 1 else "k_seed"
        dom_bytes = align_b if dom == 1 else seed_b
        achieved = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9
        cells_rank = float((lengths[pairs[mine][:, 0]] * lengths[pairs[mine][:, 1]]).sum())
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text())
                key = f"{args.workload}:{args.gpus}:{dom_name}"
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "pairwise alignments/sec", "value": total_pairs / (elapsed / args.steps), "unit": "pairs/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if args.workload == "headline" else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {num} structures x {length} residues, d={dim}, all {total_pairs} "
                                   f"pairs i<j sharded over {args.gpus} GPU(s), pipeline H (tensor-RBF SW seed -> Kabsch -> "
                                   f"coord-RBF SW score + affine DTW(1.0,0.01) -> Kabsch/RMSD/TM)",
                       "structures": num, "residues": length, "tensor_width": dim, "pairs": total_pairs,
                       "pairs_per_gpu": int(len(mine)), "seed": seed},
            "dtw_mcells_per_s": cells_rank * world / (stage_ms[1] * 1e-3) / 1e6,
            "stage_ms": {"k_seed": stage_ms[0], "k_align": stage_ms[1], "runs_averaged": runs},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": stage_ms[dom],
                         "note": "fused RBF+DP is FP64-VALU/dependency bound, not HBM bound (DESIGN.md); "
                                 "cells/s and the VALU estimate are in dtw_mcells_per_s / valu_f64"},
            "valu_f64": {"est_flop_per_cell": {"seed_fill": 59, "align_fill": 49},
                         "achieved_tflops": (59 + 49) * cells_rank / ((stage_ms[0] + stage_ms[1]) * 1e-3) / 1e12,
                         "peak_tflops": FP64_VALU_PEAK_TFLOPS},
        }
        # the binding resource, from the committed PMC summary of this workload (tools/pmc_profile.sh): share of all
        # cycles in which a SIMD issues a VALU instruction = SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs)
        try:
            if args.workload == "headline" and args.gpus == 1:
                pmc = json.load(open(ROOT / "profiles" / "r01" / "pmc_summary.json"))
                out["valu_f64"]["issue_frac_pmc"] = {
                    ("k_seed" if "k_seed" in k else "k_align"): round(v["SQ_INSTS_VALU"] / 1024 * 4 / (v["GRBM_GUI_ACTIVE"] / 8), 4)
                    for k, v in pmc.items() if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v}
        except (OSError, ValueError, KeyError):
            pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(coords, tensors, offsets, pairs, res, aln)
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        out["matrix_checksum"] = float(matrix.sum())
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
