#!/bin/bash
# Run on the GPU box from the repository root: kernel trace of tools/trace_prefill.py -> gpurun_out/prefill/kernel_stats_$QUANT.md
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prefill
mkdir -p "$OUT"
export TMPDIR=/tmp
QUANT=${QUANT:-bf16}
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$QUANT" -o p -- python3 "$ROOT/tools/trace_prefill.py" > "$OUT/stdout_$QUANT.txt"
STATS=$(find "$OUT/trace_$QUANT" -name '*kernel_stats.csv' | head -n 1)
python3 "$ROOT/tools/prof_summary.py" "$STATS" "$OUT/kernel_stats_$QUANT.md" > /dev/null
cat "$OUT/stdout_$QUANT.txt"
head -40 "$OUT/kernel_stats_$QUANT.md"
