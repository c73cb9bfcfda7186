#!/bin/bash
# Run on the GPU box from the repository root: kernel trace of tools/tune_gptq_stream.py -> per-kernel mean durations on stdout.
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_gptq
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp NO_CHECK=1
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/tune_gptq_stream.py" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "hymeta" not in n: continue
    key = n.replace("(anonymous namespace)::", "").split("(")[0][-60:] + " grid=" + r["Grid_Size_X"] if "Grid_Size_X" in r else n
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v.sort()
    print(f"{k:90s} n={len(v):5d} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}")
PY
