"""The P x P matrix alone (cr_batch_run_scores: k_seed + k_score) of the headline configuration, timed; a workload for
tools/pmc_quick.sh / tools/pmc_share.sh."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from caretta_amd import engine, synthetic
fam = synthetic.make_family(128, 300, seed=20242)
coords, tensors, offsets = synthetic.pack(fam)
pairs = engine.all_pairs(128)
ctx = engine.Context(0)
b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
prm = engine.make_params()
for _ in range(2): b.run(prm, scores_only=True)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5): b.run(prm, scores_only=True)
ctx.synchronize()
sw, _ = b.fetch_scores()
print(f"{len(pairs)} pairs of 300 x 300, matrix entries only: {(time.perf_counter()-t0)/5*1e3:.3f} ms per pass, checksum {sw.sum():.6f}")
