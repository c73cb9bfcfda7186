#!/usr/bin/env python3
"""Condense a rocprofv3 kernel_stats.csv into a short table (kernel names truncated)."""
import csv
import sys


def main(path, out=None):
    rows = list(csv.DictReader(open(path)))
    lines = ["| kernel | calls | avg us | min us | max us | total ms | % |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "")
        name = name.split("(")[0] if name.startswith(("hymeta", "void hymeta")) else name[:60] + "..."
        lines.append("| %s | %s | %.2f | %.2f | %.2f | %.3f | %.2f |" % (
            name.replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
