#!/usr/bin/env python3
"""One rank under torch.distributed.run with the nccl (= RCCL) backend: the sharded matrix through a real
all_gather_into_tensor, and a single un-warmed step whose all-gather directly follows cr_batch_run (no host
synchronisation in between) -- both against the non-distributed results, bit for bit.  Run by test_gpu_parity.py:

    CARETTA_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port P tests/rccl_single_rank_check.py
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dev = torch.device("cuda", torch.cuda.current_device())
os.environ["CARETTA_FORCE_DIST"] = "1"
dist.init_process_group("nccl", device_id=dev)
from caretta_amd import distributed as cdist, engine, synthetic  # noqa: E402

fam = synthetic.make_family(24, 200, seed=91, ragged=True, clades=3)
coords, tensors, offsets = synthetic.pack(fam)
pairs = engine.all_pairs(len(fam))
lengths = np.diff(offsets)

# (1) the library's sharded matrix: partition -> kernels -> RCCL all-gather -> P x P
m_dist = cdist.pairwise_matrix_sharded(coords, tensors, offsets, engine.make_params())

# (2) bench.py's step, first and only execution: the all-gather is queued right behind the kernels on torch's
#     current stream; if it were not ordered behind them it would gather the NaN fill
ctx = engine.Context(dev.index or 0, stream=torch.cuda.current_stream(dev).cuda_stream)
batch = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
local = torch.full((len(pairs),), float("nan"), dtype=torch.float64, device=dev)
gathered = torch.empty(len(pairs), dtype=torch.float64, device=dev)
batch.run(engine.make_params(), sw_out_device_ptr=local.data_ptr())
dist.all_gather_into_tensor(gathered, local)
got = gathered.cpu().numpy()
res, _ = batch.fetch(want_alignments=False)
batch.close()
assert not np.isnan(got).any(), "the all-gather overtook the kernels"
assert np.array_equal(got, res["sw"])
m_step = cdist.scatter_to_matrix(got.reshape(1, -1), pairs, lengths, len(fam))

# (3) the non-distributed matrix
ctx2 = engine.Context(dev.index or 0)
b2 = engine.PairBatch(ctx2, coords, tensors, offsets).set_pairs(pairs)
b2.run(engine.make_params())
sw, _ = b2.fetch_scores()
b2.close()
m_ref = engine.assemble_matrix(pairs, sw, len(fam))
assert np.array_equal(m_dist, m_ref) and np.array_equal(m_step, m_ref)
dist.barrier()
dist.destroy_process_group()
print("rccl single-rank matrix ok", m_ref.shape, "backend nccl")
