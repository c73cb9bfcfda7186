#!/usr/bin/env python3
"""HBM bytes of one whole-model decode step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

usage: step_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
Every step execution of the run (eager warm-ups and graph replays alike) launches the same kernels, 14 GLA decode launches
among them: the counters are summed over all launches of a kernel and divided by the executions that count gives.
FETCH_SIZE (KiB) is doubled (gfx950 tallies a 128-byte request of a wide coalesced read as 64 bytes); WRITE_SIZE (KiB) as read."""
import collections
import csv
import glob
import json
import os
import sys


def sums(d):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = (r.get("Kernel_Name") or r.get("Kernel Name") or "").replace("(anonymous namespace)::", "")
        name = name.split("(")[0].replace("void ", "")[:90]
        if not name.startswith("hymeta::"):
            continue          # building the synthetic model and the caches; the step's few library launches move kilobytes
        acc[name][0] += 1
        acc[name][1] += float(r["Counter_Value"])
    return acc


def main(fetch_dir, write_dir, out):
    f, w = sums(fetch_dir), sums(write_dir)
    gla = [k for k in f if "gla_decode" in k]
    execs = sum(f[k][0] for k in gla) / 14.0
    rows, total = [], 0.0
    for k in sorted(f, key=lambda k: -f[k][1]):
        rd = 2.0 * f[k][1] * 1024 / execs
        wr = w.get(k, [0, 0.0])[1] * 1024 / execs
        total += rd + wr
        rows.append({"kernel": k, "launches_per_step": round(f[k][0] / execs, 2), "read_bytes_per_step": int(rd), "write_bytes_per_step": int(wr)})
    rec = {"batch": 64, "ctx": 8192, "linears": os.environ.get("QUANT", "bf16"), "step_executions_counted": execs, "hbm_bytes_per_step": int(total),
           "correction": "FETCH_SIZE (KiB) doubled for gfx950 wide coalesced reads; WRITE_SIZE (KiB) as read",
           "kernels": [r for r in rows if r["read_bytes_per_step"] + r["write_bytes_per_step"] > (1 << 20)]}
    json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
