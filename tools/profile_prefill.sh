#!/bin/bash
# Kernel-trace statistics of the two 32k prefill kernels (run on the GPU box from the repo root).
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_prefill
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for W in swa gla; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W" -o p -- \
      python3 "$ROOT/tools/prof_one.py" $W 32768 > "$OUT/$W.txt"
  STATS=$(find "$OUT/$W" -name '*kernel_stats.csv' | head -n 1)
  python3 "$ROOT/tools/prof_summary.py" "$STATS" "$OUT/${W}_kernel_stats.md" > /dev/null
done
cat "$OUT"/*_kernel_stats.md "$OUT"/*.txt
