#!/usr/bin/env python3
"""Per-basic-block instruction histogram of a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_blocks.py file.s <substring of the kernel's mangled name> [top_n_blocks]
"""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l.split(":")[0])
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, label = [], [], "entry"
    for l in lines[start + 1:end]:
        s = l.strip()
        if re.match(r"\.LBB\d+_\d+:", s):
            blocks.append((label, cur))
            cur, label = [], s
        elif s and not s.startswith(";") and not s.startswith("."):
            cur.append(s)
    blocks.append((label, cur))
    print(lines[start].split(":")[0][:90], "- instructions:", sum(len(b) for _, b in blocks))
    for lab, b in sorted(blocks, key=lambda x: -len(x[1]))[:top]:
        c = collections.Counter(x.split()[0] for x in b)
        f64 = sum(v for k, v in c.items() if "f64" in k)
        print(f"  {lab} {len(b)} instrs, {f64} f64 ops")
        print("    " + ", ".join(f"{k}:{v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
