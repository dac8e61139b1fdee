#!/usr/bin/env python3
"""Neighbor joining wall time against the number of taxa and helper threads (CARETTA_NJ_THREADS), with the tree checked
against the single-thread result.  python tools/nj_time.py"""
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 1:
    from caretta_amd import neighbor_joining as nj
    rng = np.random.default_rng(1)
    out = []
    for p in (384, 512, 768, 1024, 1500, 2000):
        x = rng.normal(size=(p, 6))
        d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
        nj.neighbor_joining(d)
        t0 = time.perf_counter()
        tree, bl = nj.neighbor_joining(d)
        out.append(f"P={p}: {1e3 * (time.perf_counter() - t0):.1f} ms [{int(tree.sum() % 1000003)}/{float(bl.sum()):.9f}]")
    print(f"threads {os.environ.get('CARETTA_NJ_THREADS', 'auto')}: " + "  ".join(out))
else:
    for t in ("1", "2", "4", "8", "16", ""):
        env = dict(os.environ)
        if t:
            env["CARETTA_NJ_THREADS"] = t
        else:
            env.pop("CARETTA_NJ_THREADS", None)
        subprocess.run([sys.executable, __file__, "run"], env=env, check=True)
