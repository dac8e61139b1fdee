#!/usr/bin/env python3
"""The single-process multi-GPU path (cr_multi_*: one context per visible GPU, one grouped RCCL all-gather) against the
one-GPU path on BASELINE config 4 (512 x 300) or the given P x L: same scores bit for bit, and the time of both.

    python tools/multi_gpu_check.py [P [L]] [--timeout SECONDS]          prints ONE JSON line

This process is a WATCHDOG: it never touches a GPU (it imports neither torch nor the HIP library).  It starts a FRESH child
process per measurement -- the check itself, and the same again with CARETTA_MULTI_NUMA=1 (every device's host thread pinned to
the CPUs of the device's NUMA node) so that the first run on an 8-GPU node can compare them -- waits for it under a time
limit, kills exactly that child on timeout, and exits non-zero when a child failed, timed out or found a difference.
bench.py runs this whenever a one-rank run sees more than one GPU, so that a fault on hardware this path has never seen
cannot take the benchmark line with it.  (A process that has initialised the GPU is never replaced by another program.)
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def child(num, length):
    sys.path.insert(0, str(ROOT))
    import numpy as np

    from caretta_amd import engine, synthetic
    fam = synthetic.make_family(num, length, dim=10, seed=20243)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(num)
    prm = engine.make_params()
    ctx = engine.Context(0)
    reps = 3
    for it in range(reps + 1):                         # (both sides: upload, pair list, kernels, download, every call)
        if it == 1:
            t0 = time.perf_counter()
        batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        batch.run(prm, scores_only=True)
        sw_one, flags_one = batch.fetch_scores()
        batch.close()
    t_one = (time.perf_counter() - t0) / reps
    multi = engine.MultiDevice()
    for it in range(reps + 1):
        if it == 1:
            t0 = time.perf_counter()
        sw, flags = multi.pairwise_scores(coords, tensors, offsets, prm)
    t_multi = (time.perf_counter() - t0) / reps
    # the consumer of the matrix: identical guide trees (neighbor joining is 1-ulp sensitive, neighbor_joining.py:118-129)
    from caretta_amd import neighbor_joining as nj
    m_one, m_multi = engine.assemble_matrix(pairs, sw_one, num), engine.assemble_matrix(pairs, sw, num)
    tree_one, bl_one = nj.neighbor_joining(m_one.max() - m_one)
    tree_multi, bl_multi = nj.neighbor_joining(m_multi.max() - m_multi)
    trees_identical = bool(np.array_equal(tree_one, tree_multi) and np.array_equal(bl_one, bl_multi))
    out = {"devices": multi.num_devices, "structures": num, "residues": length, "pairs": int(len(pairs)),
           "one_gpu_ms": t_one * 1e3, "multi_gpu_ms": t_multi * 1e3, "speedup": t_one / t_multi,
           "last_call_ms": dict(zip(("slowest_share_events", "all_gather_events", "download_events_plus_host_scatter"), multi.last_ms())),
           "host_thread_numa_nodes": multi.numa_nodes(),
           "scores_identical": bool(np.array_equal(sw, sw_one) and np.array_equal(flags, flags_one)),
           "nj_trees_identical": trees_identical,
           "note": "one process, one context + host thread per GPU, cr_partition_pairs, one grouped ncclAllGather (RCCL bound at run "
                   "time); both times include the upload of the structures and the download of the score vector"}
    multi.close()
    print(json.dumps(out), flush=True)
    return 0 if (out["scores_identical"] and trees_identical) else 1


def run_child(args, env, limit):
    """A fresh process for one measurement; -> (record or None, failure text or None)."""
    cmd = [sys.executable, str(Path(__file__).resolve()), "--child", *args]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        proc.kill()                                    # exactly the process started above
        out, err = proc.communicate()
        return None, f"timed out after {limit:.0f} s; stderr tail: {err[-400:]}"
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    rec = json.loads(lines[-1]) if lines else None
    if proc.returncode != 0:
        return rec, f"exit code {proc.returncode}; stderr tail: {err[-400:]}"
    if rec is None:                                    # exit 0 without a record is a failure too, never an empty success
        return None, f"no JSON line from the child; stdout tail: {out[-200:]!r}; stderr tail: {err[-400:]}"
    return rec, None


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--child":
        raise SystemExit(child(int(argv[1]), int(argv[2])))
    limit = 240.0
    if "--timeout" in argv:
        k = argv.index("--timeout")
        limit = float(argv[k + 1])
        del argv[k:k + 2]
    num = argv[0] if argv else "512"
    length = argv[1] if len(argv) > 1 else "300"
    base = dict(os.environ)
    base.pop("CARETTA_MULTI_NUMA", None)
    rec, fault = run_child([num, length], base, limit)
    out = rec or {}
    failed = fault is not None
    if fault:
        out["error"] = fault
    else:
        # the same with every device's host thread on its device's NUMA node (default off: A/B on the first multi-GPU box)
        numa, numa_fault = run_child([num, length], dict(base, CARETTA_MULTI_NUMA="1"), limit)
        if numa_fault:
            out["numa_pinned"] = {"error": numa_fault}
            failed = True
        else:
            out["numa_pinned"] = {k: numa[k] for k in ("multi_gpu_ms", "speedup", "last_call_ms", "host_thread_numa_nodes", "scores_identical",
                                                      "nj_trees_identical")}
    print(json.dumps(out), flush=True)
    raise SystemExit(1 if failed else 0)


if __name__ == "__main__":
    main()
