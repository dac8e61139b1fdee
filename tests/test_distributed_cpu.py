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
