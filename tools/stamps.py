#!/usr/bin/env python3
"""In-kernel phase timing of the batch kernels (diagnostic build, never shipped).

    python tools/stamps.py build                      hipcc -DCR_STAMPS -> gpurun_out/lib/libcaretta_hip_stamps.so (scratch: never part of the tree;
                                                      `run` builds it on the GPU box when it is missing)
    python tools/stamps.py run [workload ...]         on the GPU box: medians of fill / walk / rest per kernel, in shader cycles and us

The stamped library is ONE translation unit with the default instruction scheduler, so its absolute times differ a
little from the product's; it tells which PHASE dominates a launch.  CARETTA_WIDE=R,B is honoured.
"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "caretta_amd" / "csrc"
LIB = ROOT / "gpurun_out" / "lib" / "libcaretta_hip_stamps.so"


def build():
    LIB.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-pthread", "-DCR_STAMPS",
           "-shared", str(CSRC / "cr_api.hip"), "-o", str(LIB)]
    print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(CSRC))


def run(names):
    if not LIB.exists():
        build()
    os.environ["CARETTA_HIP_LIB"] = str(LIB)
    sys.path.insert(0, str(ROOT))
    import numpy as np
    from caretta_amd import _capi, engine, synthetic
    sys.path.insert(0, str(ROOT / "tools"))
    from calibrate_wide import WORKLOADS
    lib = _capi.load()
    lib.cr_debug_stamps.restype = C.c_int
    lib.cr_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    ctx = engine.Context(0)
    prm = engine.make_params()
    for name in names:
        if name.startswith("tree"):            # progressive alignment of treeP structures of 300: the ROOT node's launches
            from caretta_amd import multiple_alignment as ma, neighbor_joining as nj
            num = int(name[4:])
            fam = synthetic.make_family(num, 300, seed=20242)
            prots = [ma.Protein(s.name, s.tensors, s.coordinates, s.sequence) for s in fam]
            msa = ma.MultipleAlignment(prots)
            p = dict(flexible=False, gamma_tensor=7.0, gamma_coords=0.03, verbose=False)
            m = msa.make_pairwise_matrix(p)
            tree, _ = nj.neighbor_joining(m.max() - m)
            msa.progressive_align(tree, 1.0, 0.01, 1.0, 1.0, p, dict(flexible=False, verbose=False))
            st = np.zeros((1, 8), dtype=np.uint64)
            _capi.check(lib.cr_debug_stamps(st.ctypes.data_as(C.c_void_p), 1))
            d = np.diff(st.astype(np.int64), axis=1)[0]
            print(f"{name} root node (width {len(next(iter(msa.final_alignments['int-final'].values()))) if hasattr(msa, 'final_alignments') else '?'}, "
                  f"CARETTA_STAGED={os.environ.get('CARETTA_STAGED', '-')}): shader-clock cycles\n"
                  f"  seed : fill {d[0]:9d}  walk {d[1]:9d}  kabsch/rest {d[2]:9d}\n"
                  f"  node : fill {d[4]:9d}  walk {d[5]:9d}  superposition/merge {d[6]:9d}", flush=True)
            continue
        num, length, seed, stride = WORKLOADS[name]
        fam = synthetic.make_family(num, length, seed=seed)
        coords, tensors, offsets = synthetic.pack(fam)
        pairs = engine.all_pairs(num)[::stride]
        b = batch_obj = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(pairs)
        for _ in range(3):
            b.run(prm)
        ctx.synchronize()
        blocks = min(len(pairs), 8192)
        st = np.zeros((blocks, 8), dtype=np.uint64)
        _capi.check(lib.cr_debug_stamps(st.ctypes.data_as(C.c_void_p), blocks))
        st = st.astype(np.int64)
        d = np.diff(st, axis=1)
        med = lambda x: float(np.median(x))
        span_seed = st[:, 3].max() - st[:, 0].min()
        span_align = st[:, 7].max() - st[:, 4].min()
        print(f"{name} ({len(pairs)} pairs of {length}, CARETTA_WIDE={os.environ.get('CARETTA_WIDE', '-')}): shader-clock cycles (s_memtime; ~2.4 GHz)\n"
              f"  seed : fill {med(d[:, 0]):9.0f}  walk {med(d[:, 1]):9.0f}  kabsch/rest {med(d[:, 2]):9.0f}   launch span {span_seed}\n"
              f"  align: fill {med(d[:, 4]):9.0f}  walk {med(d[:, 5]):9.0f}  kabsch/metrics {med(d[:, 6]):9.0f}   launch span {span_align}", flush=True)
        if getattr(lib, "cr_debug_duo_stamps", None) is not None and (os.environ.get("CARETTA_MID") != "0" or os.environ.get("CARETTA_TRIO") != "0"):
            lib.cr_debug_duo_stamps.restype = C.c_int
            lib.cr_debug_duo_stamps.argtypes = [C.c_void_p, C.c_int]
            nb = min(len(pairs), 4096)
            ds = np.zeros((nb, 4, 8), dtype=np.uint64)
            _capi.check(lib.cr_debug_duo_stamps(ds.ctypes.data_as(C.c_void_p), nb))
            ds = ds.astype(np.int64)
            if ds[:, 0, 1].any():
                t0 = st[:nb, 0]
                for wv in range(4):
                    if not ds[:, wv, 1].any():
                        continue
                    print(f"  wave {wv}: seed loop {med(ds[:, wv, 1] - ds[:, wv, 0]):9.0f} (start +{med(ds[:, wv, 0] - t0):7.0f}, waited {med(ds[:, wv, 2]):8.0f})   "
                          f"align loop {med(ds[:, wv, 5] - ds[:, wv, 4]):9.0f} (start +{med(ds[:, wv, 4] - st[:nb, 4]):7.0f}, waited {med(ds[:, wv, 6]):8.0f})", flush=True)
        tot = st[:, 7] - st[:, 0]                   # (one block's stamps share a clock: the XCDs' counters are not synchronised)
        print("  pair total (launch -> results), min p10 p50 p90 max: " + " ".join(f"{v:8.0f}" for v in np.percentile(tot, [0, 10, 50, 90, 100])), flush=True)
        if os.environ.get("STAMPS_PLACEMENT") and getattr(lib, "cr_debug_duo_stamps", None) is not None:
            # where the workgroups ran (k_pair_trio records HW_REG_HW_ID / HW_REG_XCC_ID of every wave): pair time against the
            # number of workgroups on its CU, per XCD, and against what shares wave 0's SIMD
            nb = min(len(pairs), 4096)
            ds = np.zeros((nb, 4, 8), dtype=np.uint64)
            _capi.check(lib.cr_debug_duo_stamps(ds.ctypes.data_as(C.c_void_p), nb))
            hw = ds[:, :, 3]
            if hw[:, 0].any():
                xcc, low = (hw >> np.uint64(32)).astype(np.int64) & 0xf, hw.astype(np.int64) & 0xffffffff
                simd, cu, sh, se = (low >> 4) & 3, (low >> 8) & 15, (low >> 12) & 1, (low >> 13) & 7
                cukey = (xcc[:, 0] << 12) | (se[:, 0] << 8) | (sh[:, 0] << 4) | cu[:, 0]
                tot_b = tot[:nb]
                per_cu = {}
                for k, t in zip(cukey, tot_b):
                    per_cu.setdefault(int(k), []).append(int(t))
                by_count = {}
                for k, v in per_cu.items():
                    by_count.setdefault(len(v), []).extend(v)
                print("  workgroups per CU -> CUs, median pair total: " + "; ".join(f"{c}: {sum(1 for v in per_cu.values() if len(v) == c)} CUs, {np.median(by_count[c]):.0f}" for c in sorted(by_count)), flush=True)
                print("  per XCD: launch ramp (last - first start), finish spread (last - first end), first start -> last end: " + "; ".join(
                    f"{x}: {int(st[:nb, 0][xcc[:, 0] == x].max() - st[:nb, 0][xcc[:, 0] == x].min())}, {int(st[:nb, 7][xcc[:, 0] == x].max() - st[:nb, 7][xcc[:, 0] == x].min())}, "
                    f"{int(st[:nb, 7][xcc[:, 0] == x].max() - st[:nb, 0][xcc[:, 0] == x].min())}" for x in range(8) if (xcc[:, 0] == x).any()), flush=True)
                print("  per XCD (workgroups, median pair total): " + "; ".join(f"{x}: {int((xcc[:, 0] == x).sum())}, {np.median(tot_b[xcc[:, 0] == x]):.0f}" for x in range(8) if (xcc[:, 0] == x).any()), flush=True)
                waves_used = int((hw != 0).any(axis=0).sum())
                simd_key = lambda wv: (cukey << 2) | simd[:, wv]
                cons_per_simd = {}
                for k in simd_key(0):
                    cons_per_simd[int(k)] = cons_per_simd.get(int(k), 0) + 1
                all_per_simd = {}
                for wv in range(waves_used):
                    for k in simd_key(wv):
                        all_per_simd[int(k)] = all_per_simd.get(int(k), 0) + 1
                g = {}
                for b in range(nb):
                    key = (cons_per_simd[int(simd_key(0)[b])], all_per_simd[int(simd_key(0)[b])])
                    g.setdefault(key, []).append(int(tot_b[b]))
                print("  (recurrence waves, all waves) on wave 0's SIMD -> pairs, median pair total: " + "; ".join(f"{k}: {len(v)}, {np.median(v):.0f}" for k, v in sorted(g.items())), flush=True)
                order = np.argsort(tot_b)
                fast, slow = order[:nb // 10], order[-nb // 10:]
                print(f"  fastest tenth: block ids min/median/max {fast.min()} {int(np.median(fast))} {fast.max()}; slowest tenth: {slow.min()} {int(np.median(slow))} {slow.max()}", flush=True)
        if os.environ.get("STAMPS_DETAIL"):
            pc = lambda x: " ".join(f"{v:8.0f}" for v in np.percentile(x, [0, 10, 50, 90, 100]))
            t00 = st[:, 0].min()
            print(f"  percentiles (min p10 p50 p90 max): seed start {pc(st[:, 0] - t00)} | seed end {pc(st[:, 3] - t00)}\n"
                  f"      align start {pc(st[:, 4] - t00)} | align end {pc(st[:, 7] - t00)}\n"
                  f"      seed fill {pc(d[:, 0])} | align fill {pc(d[:, 4])}", flush=True)
        batch_obj.close()


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("build", "run"):
        raise SystemExit(__doc__)
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] or ["c5share", "c2"])
