#!/usr/bin/env python3
"""Does the process exit cleanly when torch's librccl.so is loaded BY PATH before `import torch`?  (It does not on ROCm 7.0 /
torch 2.10: "double free or corruption" in the teardown -- which is why caretta_amd._capi.share_torch_rccl imports torch.)

    python tools/exit_order_probe.py ctx_then_torch | rccl_by_path_then_torch | share_then_torch
"""
import ctypes
import importlib.util
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caretta_amd import engine, synthetic, _capi
mode = sys.argv[1]
fam = synthetic.make_family(24, 150, seed=1)
coords, tensors, offsets = synthetic.pack(fam)
prm = engine.make_params()
if mode == "rccl_by_path_then_torch":
    ctypes.CDLL(os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib", "librccl.so"), mode=ctypes.RTLD_GLOBAL)
if mode == "share_then_torch":
    _capi.share_torch_rccl()
ctx = engine.Context(0)
b = engine.PairBatch(ctx, coords, tensors, offsets).set_pairs(engine.all_pairs(24))
b.run(prm); b.fetch(); b.close(); ctx.close()
import torch
print("device", torch.cuda.current_device(), mode, "done")
