"""All-vs-all pair set sharded over the GPUs of one node.

Pairs are independent (multiple_alignment.py:162-169 has no cross-pair dependency), so each rank
runs the pipeline on its share with no data-path collective; ONE all-gather of the per-rank score
vectors (RCCL over xGMI under the ``nccl`` backend, gloo in CPU tests) assembles the P x P matrix
that neighbor joining consumes.  The result is independent of the number of ranks bit for bit.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


def partition_pairs(pairs: np.ndarray, lengths: np.ndarray, world: int, rank: int) -> np.ndarray:
    """Indices (into ``pairs``) owned by ``rank``: sort by DP cell count (descending, stable on the
    pair id) and deal round-robin, so every rank gets the same mix of costs.  For equal-length
    structures this is simply ``p % world``."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    lengths = np.asarray(lengths, dtype=np.int64)
    cost = lengths[pairs[:, 0]] * lengths[pairs[:, 1]]
    order = np.argsort(-cost, kind="stable")
    return np.sort(order[rank::world])      # (cr_partition_pairs is the same deal for the all-pairs list: test_capi_cpu)


def shard_size(npairs: int, world: int) -> int:
    return (npairs + world - 1) // world


def gather_scores(local_scores, world: int, group=None):
    """All-gather equal-length per-rank score vectors (torch tensors, padded with NaN)."""
    import os

    import torch
    import torch.distributed as dist
    # CARETTA_FORCE_DIST=1: take the collective also with a single rank (exercises RCCL on a one-GPU box)
    forced = dist.is_initialized() and os.environ.get("CARETTA_FORCE_DIST") == "1"
    if world == 1 and not forced:
        return local_scores.unsqueeze(0)
    out = torch.empty(world * local_scores.numel(), dtype=local_scores.dtype, device=local_scores.device)
    dist.all_gather_into_tensor(out, local_scores.contiguous(), group=group)
    return out.view(world, local_scores.numel())


def scatter_to_matrix(gathered: np.ndarray, pairs: np.ndarray, lengths: np.ndarray, num: int) -> np.ndarray:
    """(world, shard) gathered scores -> symmetric P x P matrix (zero diagonal)."""
    world = gathered.shape[0]
    scores = np.full(len(pairs), np.nan)
    for r in range(world):
        idx = partition_pairs(pairs, lengths, world, r)
        scores[idx] = gathered[r, :len(idx)]
    if np.isnan(scores).any():
        raise RuntimeError("all-gather left pairs without a score")
    from .engine import assemble_matrix
    return assemble_matrix(pairs, scores, num)


def pairwise_matrix_sharded(coords, tensors, offsets, params=None, group=None,
                            compute_fn: Optional[Callable] = None, device=None) -> np.ndarray:
    """P x P smith_waterman_score matrix (multiple_alignment.py:158-170) with the pair set sharded
    over the ranks of ``group``.  ``compute_fn(coords, tensors, offsets, pairs) -> scores`` replaces the
    HIP engine in CPU tests (the product default fails loudly without a GPU)."""
    import torch
    import torch.distributed as dist
    from .engine import all_pairs
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    offsets = np.asarray(offsets, dtype=np.int64)
    lengths = np.diff(offsets)
    num = len(lengths)
    pairs = all_pairs(num)
    mine = partition_pairs(pairs, lengths, world, rank)
    size = shard_size(len(pairs), world)
    bad_local = 0
    if compute_fn is not None:
        local = torch.full((size,), float("nan"), dtype=torch.float64)
        local[:len(mine)] = torch.from_numpy(np.asarray(compute_fn(coords, tensors, offsets, pairs[mine]), dtype=np.float64))
    else:
        from . import _capi
        from .engine import Context, PairBatch, make_params
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # the kernels run on torch's current stream (also when that is the legacy default stream, handle 0): the fill
        # of `local` before them and the all-gather after them are ordered with them by the stream itself
        ctx = Context(dev.index or 0, stream=torch.cuda.current_stream(dev).cuda_stream)
        local = torch.full((size,), float("nan"), dtype=torch.float64, device=dev)
        batch = PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs[mine])
        if params is None:       # Protein.score_function's defaults (multiple_alignment.py:321-322), as make_pairwise_matrix
            params = make_params(gamma_tensor=0.03, gamma_coords=0.03)
        batch.run(params, sw_out_device_ptr=local.data_ptr(), scores_only=True)      # the matrix needs no alignments
        _, flags = batch.fetch_scores()
        batch.close()
        ctx.close()
        # The reference raises at a pair whose tensor score matrix has no positive local alignment (smith_waterman:
        # max_pos is None).  That condition travels as its own count (below), not as a NaN among the scores: a NaN in
        # the gathered vector then only ever means a collective or partition fault and is reported as such.
        bad_local = int(np.count_nonzero(flags & _capi.FLAG_SEED_ALL_ZERO))
    # every rank takes part in both collectives and every rank raises the same exception afterwards
    bad = torch.tensor([float(bad_local)], dtype=torch.float64, device=local.device)
    if dist.is_initialized() and world > 1:
        dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=group)
    gathered = gather_scores(local, world, group)
    if float(bad.item()) > 0:
        raise TypeError("a pair of the family has no positive local alignment of its tensor score matrix "
                        "(reference: max_pos is None)")
    return scatter_to_matrix(gathered.cpu().numpy(), pairs, lengths, num)
