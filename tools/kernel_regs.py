#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of the gfx950 kernels in libcaretta_hip.so (code-object metadata).

    python tools/kernel_regs.py [substring of the demangled kernel name ...]
"""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")


def main():
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        lib = Path(tmp) / "lib.so"
        shutil.copy(ROOT / "caretta_amd" / "csrc" / "libcaretta_hip.so", lib)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp)
        for co in sorted(Path(tmp).glob("lib.so.*gfx950")):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            for block in notes.split("  - .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*", "", dem).replace("void cr::", "")
                if pats and not any(p in dem for p in pats):
                    continue
                get = lambda key: (re.search(rf"\.{key}:\s+(\d+)", block) or [0, "?"])[1]
                print(f"{dem:60s} vgpr {get('vgpr_count'):>4s} sgpr {get('sgpr_count'):>4s} scratch {get('private_segment_fixed_size'):>5s} "
                      f"spill v{get('vgpr_spill_count')} s{get('sgpr_spill_count')} lds {get('group_segment_fixed_size')}")


if __name__ == "__main__":
    main()
