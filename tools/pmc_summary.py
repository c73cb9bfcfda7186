#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter_collection.csv per (kernel, counter).

usage: pmc_summary.py <counter_collection.csv> [kernel-substring]
Prints one line per kernel/counter: name, counter, launches, mean value per launch."""
import collections
import csv
import json
import sys


def summarise(path, needle=None):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        if needle and needle not in name:
            continue
        key = (name.split("(")[0].replace("void ", "")[:80], r["Counter_Name"])
        acc[key][0] += 1
        acc[key][1] += float(r["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


if __name__ == "__main__":
    out = summarise(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    for (name, ctr), (n, mean) in sorted(out.items()):
        print(json.dumps({"kernel": name, "counter": ctr, "launches": n, "mean_per_launch": mean}))
