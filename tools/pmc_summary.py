#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (one directory per pass) into per-kernel means per dispatch."""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0].replace("cr::", "")


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
        for (did, counter), value in per_dispatch.items():
            acc[names[did]][counter].append(value)
    out = {}
    for kernel, counters in sorted(acc.items()):
        if not kernel.startswith("k_"):
            continue
        out[kernel] = {c: sum(v) / len(v) for c, v in sorted(counters.items())}
        out[kernel]["dispatches"] = len(next(iter(counters.values())))
    print(json.dumps(out, indent=1))
    with open(f"{root}/summary.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
