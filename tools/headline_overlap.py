"""Does the headline (128 x 300, all 8128 pairs) gain from running as SEVERAL lists on several streams, so that one list's
alignment kernel fills the tail of another's seed kernel?  One batch in one context against K batches in K contexts (private
streams), all enqueued before the first wait.   python tools/headline_overlap.py"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from caretta_amd import engine, synthetic

fam = synthetic.make_family(128, 300, seed=20242)
coords, tensors, offsets = synthetic.pack(fam)
pairs = engine.all_pairs(128)
prm = engine.make_params(gamma_tensor=7.0, gamma_coords=0.03, gap_open=1.0, gap_extend=0.01)


def timed(splits, reps=30):
    ctxs = [engine.Context(0) for _ in splits]
    bounds = np.concatenate([[0], np.cumsum(splits)])
    batches = [engine.PairBatch(c, coords, tensors, offsets).set_pairs(pairs[bounds[k]:bounds[k + 1]]) for k, c in enumerate(ctxs)]
    for _ in range(3):
        for b in batches:
            b.run(prm)
        for c in ctxs:
            c.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for b in batches:
            b.run(prm)
        for c in ctxs:
            c.synchronize()
        ts.append(time.perf_counter() - t0)
    sw = np.concatenate([b.fetch(want_alignments=False)[0]["sw"] for b in batches])
    for b in batches:
        b.close()
    ts = np.sort(ts) * 1e3
    return ts[0], float(np.median(ts)), float(sw.sum())


n = len(pairs)
for splits in ([n], [n // 2, n - n // 2], [n // 3, n // 3, n - 2 * (n // 3)], [n // 4] * 3 + [n - 3 * (n // 4)], [2032, n - 2032], [3072, n - 3072],
               [3072, 3072, n - 6144], [1024, 3072, n - 4096], [n]):
    lo, med, chk = timed(splits)
    print(f"{str(splits):34s}: min {lo:.3f} median {med:.3f} ms   (checksum {chk:.6f})")
