"""World-size-2 test of the sharded all-vs-all path on CPU (gloo).  The per-rank compute is injected
(the C oracle stands in for the HIP engine, which refuses to run without a GPU); what is tested is the
product's partitioning, the all-gather and the matrix assembly: the result must not depend on the
number of ranks, bit for bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_scores(coords, tensors, offsets, pairs):
    from oracle.pyoracle import Oracle
    outs, _ = Oracle().pairwise_batch(coords, tensors, offsets, pairs, want_aln=False)
    return outs["sw"]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from caretta_amd import distributed as cdist
    from caretta_amd import synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fam = synthetic.make_family(7, 40, seed=5, ragged=True)
    coords, tensors, offsets = synthetic.pack(fam)
    m = cdist.pairwise_matrix_sharded(coords, tensors, offsets, compute_fn=_oracle_scores)
    np.save(Path(out_dir) / f"m{rank}.npy", m)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_all_pairs_once():
    from caretta_amd import distributed as cdist
    from caretta_amd import engine
    rng = np.random.default_rng(0)
    lengths = rng.integers(30, 300, size=19)
    pairs = engine.all_pairs(19)
    for world in (1, 2, 3, 8):
        parts = [cdist.partition_pairs(pairs, lengths, world, r) for r in range(world)]
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(len(pairs)))
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) <= cdist.shard_size(len(pairs), world)
        cost = lengths[pairs[:, 0]] * lengths[pairs[:, 1]]
        loads = [cost[p].sum() for p in parts]
        assert max(loads) <= 1.15 * min(loads)
    eq = cdist.partition_pairs(pairs, np.full(19, 100), 4, 1)
    assert np.array_equal(eq, np.arange(1, len(pairs), 4))          # equal lengths: p % world


def test_world2_gloo_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = np.load(tmp_path / "m0.npy"), np.load(tmp_path / "m1.npy")
    assert np.array_equal(m0, m1)
    from caretta_amd import engine, synthetic
    fam = synthetic.make_family(7, 40, seed=5, ragged=True)
    coords, tensors, offsets = synthetic.pack(fam)
    pairs = engine.all_pairs(7)
    ref = engine.assemble_matrix(pairs, _oracle_scores(coords, tensors, offsets, pairs), 7)
    assert np.array_equal(m0, ref)                                  # independent of the rank count, bit for bit
    assert np.array_equal(m0, m0.T) and np.all(np.diag(m0) == 0)


def test_default_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from caretta_amd import distributed as cdist
    from caretta_amd import synthetic
    fam = synthetic.make_family(3, 20, seed=1)
    with pytest.raises(Exception):
        cdist.pairwise_matrix_sharded(*synthetic.pack(fam))


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two rank processes itself.  Without a GPU each
    RANK (not the launcher) stops with the "needs an MI355X" message, and the parent reports their exit codes."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = env["ROCR_VISIBLE_DEVICES"] = ""          # also on a GPU box: no device for the ranks
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "must be launched with" not in p.stderr
    assert p.stderr.count("needs an MI355X") == 2, p.stderr
    assert "rank exit codes [1, 1]" in p.stderr


# ---------------------------------------------------------------------------------------------------------------------
# The gates of an N-rank bench line (tools/bench_lib.py), world size 2 over gloo with the per-pair compute injected: the
# deal, the all-gather, `multi_gpu_gate` (gathered vector = one-rank vector bit for bit, identical neighbor-joining trees),
# the rank records and the verdict every rank exits with are the code `bench.py --gpus N` runs.
# ---------------------------------------------------------------------------------------------------------------------
def _fake_scores(corrupt_rank=None):
    """A deterministic score per pair from the pair's ids (any function of the pair alone will do: the gate compares two
    ways of computing it).  corrupt_rank: that rank flips the last bit of its first score -- the fault the gate must catch."""
    def compute(coords, tensors, offsets, pairs):
        p = np.asarray(pairs, dtype=np.float64)
        s = 1.0 + np.sin(p[:, 0] * 12.9898 + p[:, 1] * 78.233) ** 2 + p[:, 0] / 64.0
        import torch.distributed as dist
        if corrupt_rank is not None and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_rank() == corrupt_rank:
            s = s.copy()
            s[:1] = np.nextafter(s[:1], np.inf)
        return s
    return compute


def _gate_worker(rank, world, port, out_dir, corrupt_rank):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tools"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import json

    import torch.distributed as dist
    import bench_lib as bl
    from caretta_amd import synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt = bl.CpuRuntime(world, rank, _fake_scores(corrupt_rank))
    fam = synthetic.make_family(12, 30, seed=3)
    rec = bl.multi_gpu_record(rt, "test", steps=1, warmup=0, family=fam, cfg=(12, 30, 3))
    info = bl.rank_records(rt)
    failed = bl.failed_gates({"test_sharded": rec}) if rank == 0 else []
    bad = rt.broadcast_flag(bool(failed))
    (Path(out_dir) / f"verdict{rank}.json").write_text(json.dumps({"bad": bad, "failed": failed, "rec": rec, "ranks": info if rank == 0 else None}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt_rank", [None, 1])
def test_bench_multi_gpu_gate_world2_gloo(tmp_path, corrupt_rank):
    import json

    import torch.multiprocessing as mp
    mp.spawn(_gate_worker, args=(2, _free_port(), str(tmp_path), corrupt_rank), nprocs=2, join=True)
    v0, v1 = (json.loads((tmp_path / f"verdict{r}.json").read_text()) for r in (0, 1))
    rec = v0["rec"]
    assert v1["rec"] is None                                         # only rank 0 holds the record
    gate = rec["multi_gpu_gate"]
    assert rec["n_gpus"] == 2 and rec["pairs"] == 66 and rec["all_gather_ms"] is not None
    assert [r["rank"] for r in v0["ranks"]["ranks"]] == [0, 1] and v0["ranks"]["collective"]["backend"] == "gloo"
    assert len({r["pid"] for r in v0["ranks"]["ranks"]}) == 2
    if corrupt_rank is None:
        assert gate == {"pairs": 66, "matrix_equal": True, "nan_scores": 0, "differing_pairs": 0, "trees_identical": True, "tree_rows": 21}
        assert v0["failed"] == [] and v0["bad"] is False and v1["bad"] is False
    else:
        # ONE score of rank 1 one ulp off: reported, and EVERY rank learns the verdict (bench.py exits non-zero on all of them)
        assert gate["matrix_equal"] is False and gate["differing_pairs"] == 1
        assert "test_sharded.multi_gpu_gate.matrix_equal" in v0["failed"]
        assert v0["bad"] is True and v1["bad"] is True


def test_bench_gate_walkers():
    sys.path.insert(0, str(ROOT / "tools"))
    import bench_lib as bl
    ok = {"a": {"matrix_equal": True, "trees_identical": True, "pair_gate": {"mismatches": 0}}, "cpu_baseline": {"parity_mismatches": 0}}
    assert bl.gate_ok(ok) and bl.failed_gates(ok) == []
    bad = {"c4_sharded": {"multi_gpu_gate": {"matrix_equal": True, "trees_identical": False}, "pair_gate_own_share": {"mismatches": 2}},
           "list": [{"bipartitions_equal": False}]}
    assert not bl.gate_ok(bad)
    assert bl.failed_gates(bad) == ["c4_sharded.multi_gpu_gate.trees_identical", "c4_sharded.pair_gate_own_share.mismatches=2",
                                    "list[0].bipartitions_equal"]
    # a NaN among the gathered scores can never pass as equal
    from caretta_amd import engine
    pairs = engine.all_pairs(4)
    s = np.arange(6, dtype=np.float64) + 1
    t = s.copy()
    t[2] = np.nan
    g = bl.multi_gpu_gate(pairs, 4, t, s)
    assert g["matrix_equal"] is False and g["nan_scores"] == 1 and g["trees_identical"] is False
    assert bl.multi_gpu_gate(pairs, 4, s, s.copy())["matrix_equal"] is True
    # the compact table that closes the line
    extras = {"c3_sharded": {"n_gpus": 1, "ms": 3.0, "share_of_2": {"ms": 1.6, "projected_speedup_2gpu": 1.9}, "share_of_4": {"ms": 0.9, "projected_speedup_4gpu": 3.3},
                             "share_of_8": {"ms": 0.5, "projected_speedup_8gpu": 6.0}}}
    assert bl.shares_summary(extras, 1)["c3"] == {"ms_1gpu": 3.0, "2": [1.6, 1.9], "4": [0.9, 3.3], "8": [0.5, 6.0]}
    extras = {"c5_sharded": {"n_gpus": 8, "ms": 2.0, "ms_1gpu": 13.0, "speedup_vs_1gpu": 6.5, "all_gather_ms": 0.03,
                             "multi_gpu_gate": {"matrix_equal": True, "trees_identical": True}, "pair_gate_own_share": {"mismatches": 0}}}
    assert bl.shares_summary(extras, 8)["c5"] == {"ms_1gpu": 13.0, "measured_8": [2.0, 6.5, 0.03, True, True, 0]}
