#!/usr/bin/env python3
"""Benchmark of the all-vs-all pairwise structural alignment path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: starts its own N rank processes, one per GPU; under
                                                          torch.distributed.run it is one of the ranks already)

A "step" is one pass of the pairwise pipeline (seed fill -> seed traceback+Kabsch -> alignment fill
-> alignment traceback+metrics, plus the all-gather of the score vectors when N>1) over the whole
pair set of a synthetic family whose structures are already resident in HBM.  Prints ONE JSON line.

`value` at EVERY N is BASELINE config 3 -- 128 structures x 300 residues, all 8 128 pairs, d = 10, seed 20242 -- with the pair
set FIXED and sharded over the N ranks (strong scaling, as the north star words it: SCALE's N = 1 is BENCH's line).  The
weak-scaling companion (P = round(128 sqrt(N)): per-GPU work fixed) is `value_weak`.  Other workloads (--workload): c2 32 x 150,
c4 512 x 300, c5 64 x 1200, each FIXED and sharded.

Besides `value` (device-resident rate, SURVEY.md 8(d) and the driver's contract; the timed region of K steps is REPEATED
`--repeats` times and `ms_per_step` / `value` are the MEDIAN run's, min / median / max in `repeats`) the line carries
  value_incl_transfers : the same pair set INCLUDING the upload of the structures and the download of every result, one batch with
                         a wait behind it -- the metric as SURVEY.md 8(d) words it; ..._pipelined: a two-stream pipeline of them;
  c3_sharded, c4_sharded, c5_sharded : BASELINE configs 3, 4 and 5 with the pair set FIXED.  N = 1: the config on this GPU and one
                         GPU's share of the 2-, 4-, 8-GPU split run alone (`share_of_G`), gated against the oracle (`nj_gate`,
                         `pair_gate`).  N > 1 (or CARETTA_FORCE_DIST=1): sharded over the N ranks + one all-gather, the
                         all-gather's own time, the speed-up against ONE GPU running the whole config (rank 0), and
                         `multi_gpu_gate`: the gathered score vector equals the one-GPU vector BIT FOR BIT and the
                         neighbor-joining trees of both are identical; `pair_gate_own_share`: rank 0's share against the oracle;
  ranks                : (N > 1) rank -> device, PCI bus id, pid of every rank (all_gather_object), the RCCL version;
  matrix_only, msa, ragged, mixed, explicit_batch : the matrix entries alone; neighbor joining + progressive alignment behind
                         the matrix (every node replayed by the oracle); ragged / mixed-size families; the reference's
                         functions on explicit score matrices with their tracebacks (tools/bench_lib.py, ragged_time.py,
                         explicit_batch_rate.py);
  roofline, cpu_baseline : as the contract asks (roofline.frac from SURVEY 8(d)'s algorithmic bytes);
  shares               : LAST key -- the compact strong-scaling table of c3 / c4 / c5 (tools/bench_lib.shares_summary).
A failed gate anywhere in the line makes EVERY rank exit non-zero (after the line is printed, with `gates_failed` in it).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


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
    ap.add_argument("--no-extras", action="store_true", help="the contract's line only: no transfers / sharded configs / consumers")
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
    # CARETTA_FORCE_DIST=1 runs EVERY branch of an N-rank run (process group over RCCL, all-gather, barrier, max-reduce, the
    # N-ranks-against-one-GPU gate, the rank records) with a single rank, so that it can be exercised on a 1-GPU box
    use_dist = world > 1 or (os.environ.get("CARETTA_FORCE_DIST") == "1" and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if use_dist:
        dist.barrier()

    import bench_lib as bl
    from caretta_amd import engine

    rt = bl.GpuRuntime(world, rank, local_rank, use_dist)
    dim, ctx, params = rt.dim, rt.ctx, rt.params

    # ------------------------------------------------------------------ the headline, timed as the contract says
    num, length, seed = bl.workload(args.workload)
    head = bl.Sharded(rt, num, length, seed)
    for _ in range(args.warmup):
        head.step()
    rt.fence()
    repeats = max(1, args.repeats)
    ctx.set_profiling(min(args.steps * repeats, 4096))
    run_s = []
    for _ in range(repeats):
        # one timed region as the contract words it: EXACTLY `steps` steps between barrier + synchronize, max over ranks
        rt.fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            head.step()
        rt.fence()
        run_s.append(rt.max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(run_s))                  # the median run's time is the line's ms_per_step / value
    stage_ms, runs = head.batch.stage_ms()
    ctx.set_profiling(0)
    res, aln = head.batch.fetch(want_alignments=(rank == 0))
    pairs, lengths, mine = head.pairs, head.lengths, head.mine
    coords, tensors, offsets = head.coords, head.tensors, head.offsets
    matrix = None
    if rank == 0:
        head_scores = head.scores()
        if np.isnan(head_scores).any():
            raise SystemExit(f"bench.py: the all-gather left {int(np.isnan(head_scores).sum())} pairs without a score")
        matrix = engine.assemble_matrix(pairs, head_scores, num)

    extras = {}
    gated = not args.no_cpu_baseline
    orc = None
    if gated and rank == 0:
        from oracle.pyoracle import Oracle
        orc = Oracle()
    if not args.no_extras:
        # ---------------------------------------------------------- the P x P matrix alone (no pairwise alignments)
        t_mat = head.time(max(3, min(args.steps, 10)), 2, scores_only=True)
        if rank == 0:
            extras["matrix_only"] = {"ms_per_step": t_mat * 1e3, "pairs_per_s": len(pairs) / t_mat,
                                     "note": "cr_batch_run_scores: seed kernel + smith_waterman_score of the coordinate score matrix per pair "
                                             "(multiple_alignment.py:158-170), no dtw_align / traceback / metrics"}
        rec = bl.incl_transfers_record(rt, head, res, aln, elapsed / args.steps, args.steps)
        if rank == 0:
            extras.update(rec)
        if args.workload != "c2":
            rec = bl.c2_record(rt, orc)
            if rank == 0:
                extras["c2"] = rec
        # ---------------------------------------------------------- BASELINE configs 3, 4 and 5 with the pair set FIXED
        for key in ("c3", "c4", "c5"):
            rec = {}
            if world == 1 and rank == 0:
                rec = bl.shares_record(rt, key, orc)
            if use_dist:
                # sharded over the ranks + one all-gather; rank 0 also runs the whole config on ONE GPU: speed-up, and the
                # gathered scores against the one-GPU scores bit for bit + identical neighbor-joining trees
                multi = bl.multi_gpu_record(rt, key, steps=10 if key == "c3" else 5, warmup=2, orc=orc)
                if rank == 0:
                    for k in ("multi_gpu_gate", "pair_gate_own_share", "all_gather_ms", "layout_of_share"):
                        if k in multi:
                            rec[k] = multi[k]
                    if world > 1:
                        rec.update({k: v for k, v in multi.items() if k not in rec})
                    else:
                        rec["forced_dist_ms"] = multi["ms"]
            if rank == 0:
                extras[f"{key}_sharded"] = rec
        if use_dist:
            info = bl.rank_records(rt)
            if rank == 0:
                extras["ranks"] = info
        # ---------------------------------------------------------- the weak-scaling companion of the headline
        if args.workload == "headline" and world > 1:
            wn, wl, ws = bl.weak_workload(world)
            weak = bl.Sharded(rt, wn, wl, ws)
            t_weak = weak.time(args.steps, args.warmup)
            if rank == 0:
                extras["value_weak"] = len(weak.pairs) / t_weak
                extras["weak"] = {"structures": wn, "residues": wl, "pairs": int(len(weak.pairs)), "ms_per_step": t_weak * 1e3,
                                  "note": "P = round(128 sqrt(N)) structures of 300: per-GPU work fixed as N grows (NOT a BASELINE config at N > 1)"}
            weak.close()

    if rank == 0 and world == 1 and not args.no_extras and args.workload == "headline":
        # ---------------------------------------------------------- the consumers behind the matrix
        extras["msa"] = bl.msa_record(rt, num, length, seed, matrix, orc)
        # ---------------------------------------------------------- what the reference actually runs: ragged families, and its
        # functions on explicit score matrices WITH their tracebacks (dynamic_time_warping.py:148-184, :226-278)
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
            fam_rec = explicit_batch_rate.explicit_record(8128, 300, matrices="family")
            extras["explicit_batch"]["family_matrices"] = {k: v for k, v in fam_rec["functions"].items() if "WITH" in k}
            extras["explicit_batch"]["family_matrices"]["what"] = ("the same list with score matrices as the reference forms them (tensor RBF of two related "
                                                                   "structures): a walk through random scores with free gaps is the worst case (runs of 1-2 cells)")
            tfile = bl.latest_profile("explicit_batch_pmc.json")
            if tfile is not None:
                extras["explicit_batch"]["traffic_over_algorithmic"] = json.loads(tfile.read_text()).get("traffic_over_algorithmic")
                extras["explicit_batch"]["traffic_source"] = str(tfile.relative_to(ROOT)) + " (a committed PMC pass, not measured in this run)"
        except Exception as exc:                          # noqa: BLE001
            extras["explicit_batch"] = {"error": repr(exc)}

    if rank == 0 and world == 1 and not args.no_extras and engine.device_count() > 1:
        # several GPUs visible to this ONE process: the single-process multi-GPU path behind make_pairwise_matrix
        # (cr_multi_*) on BASELINE config 4, in a child process with a time limit (tools/multi_gpu_check.py)
        import subprocess
        try:
            p = subprocess.run([sys.executable, str(ROOT / "tools" / "multi_gpu_check.py"), "--timeout", "240"], capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            rec_m = json.loads(line[-1]) if line else {}
            if p.returncode != 0:
                rec_m.setdefault("error", f"exit code {p.returncode}")
                rec_m["stderr_tail"] = p.stderr[-500:]
            extras["single_process_multi_gpu"] = rec_m
        except Exception as exc:                          # noqa: BLE001 -- the headline line must survive
            extras["single_process_multi_gpu"] = {"error": repr(exc)}

    failed = []
    if rank == 0:
        total_pairs = len(pairs)
        ms_per_step = elapsed / args.steps * 1e3
        sb = bl.stage_bytes(lengths, pairs[mine], dim)
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
            except Exception:                             # noqa: BLE001
                traffic = None
        out = {
            "metric": "pairwise alignments/sec", "value": total_pairs / (elapsed / args.steps), "unit": "pairs/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": {"runs": repeats, "steps_per_run": args.steps, "ms_per_step_min": min(run_s) / args.steps * 1e3,
                        "ms_per_step_median": elapsed / args.steps * 1e3, "ms_per_step_max": max(run_s) / args.steps * 1e3,
                        "note": "the timed region (K steps between barrier + synchronize) repeated; ms_per_step / value are the median run's"},
            "config": {"workload": f"{args.workload}: {num} structures x {length} residues, d={dim}, all {total_pairs} "
                                   f"pairs i<j FIXED and sharded over {args.gpus} GPU(s) + one all-gather of the scores, pipeline H "
                                   f"(tensor-RBF SW seed -> Kabsch -> coord-RBF SW score + affine DTW(1.0,0.01) -> Kabsch/RMSD/TM)",
                       "structures": num, "residues": length, "tensor_width": dim, "pairs": total_pairs,
                       "pairs_per_gpu": int(len(mine)), "seed": seed},
            "dtw_mcells_per_s": cells_rank * world / (stage_ms[1] * 1e-3) / 1e6 if stage_ms[1] > 0 else None,
            "stage_ms": {"k_seed": stage_ms[0], "k_align": stage_ms[1], "runs_averaged": runs},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": bl.HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / bl.HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": stage_ms[dom],
                         "frac_with_readback": achieved_rb / bl.HBM_PEAK_GBS,
                         "algorithmic_bytes_with_readback": sb[dom_name + "_readback"],
                         "note": "SURVEY 8(d) bytes: features in + packed decisions out.  The fused RBF+DP kernels are bound by "
                                 "FP64-rate VALU issue, not HBM (DESIGN.md section 5; valu_f64 below); the HBM-bound kernels of the "
                                 "path are the batched explicit-matrix sweeps (explicit_batch)"},
            "valu_f64": {"est_flop_per_cell": {"seed_fill": 59, "align_fill": 49},
                         "achieved_tflops": (59 + 49) * cells_rank / ((stage_ms[0] + stage_ms[1]) * 1e-3) / 1e12,
                         "peak_tflops": bl.FP64_VALU_PEAK_TFLOPS},
        }
        # the binding resource, from the newest committed PMC summary of this workload (tools/pmc_profile.sh): share of all
        # cycles in which a SIMD issues a VALU instruction = SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs)
        try:
            f = bl.latest_profile("pmc_summary.json") if (args.workload in ("headline", "c3") and args.gpus == 1) else None
            if f is not None:
                pmc = json.load(open(f))
                out["valu_f64"]["issue_frac_pmc"] = {
                    k.split("<")[0]: round(v["SQ_INSTS_VALU"] / 1024 * 4 / (v["GRBM_GUI_ACTIVE"] / 8), 4)
                    for k, v in pmc.items()
                    if k.split("<")[0] in ("k_seed", "k_align") and "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v}
                out["valu_f64"]["issue_frac_pmc"]["source"] = str(f.relative_to(ROOT)) + " (a committed PMC pass, not measured in this run)"
        except (OSError, ValueError, KeyError):
            pass
        out.update(extras)
        out.setdefault("value_weak", out["value"])         # (N = 1: the weak and the strong headline are the same job)
        out["build"] = ge.build_provenance()
        if world == 1 and gated:
            out["cpu_baseline"], out["nj_gate"] = bl.cpu_baseline(coords, tensors, offsets, pairs, res, aln, matrix)
            out["speedup_vs_cpu_1thread"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        out["matrix_checksum"] = float(matrix.sum())
        failed = bl.failed_gates(out)
        out["gates_failed"] = failed
        if not args.no_extras:
            out["shares"] = bl.shares_summary(extras, world)          # LAST key: survives the tail of a truncated record
        print(json.dumps(out))
        sys.stdout.flush()
    head.close()
    bad = rt.broadcast_flag(bool(failed))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        if rank == 0:
            print(f"bench.py: FAILED gates: {failed}", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
