#!/bin/bash
# Kernel-trace statistics of the whole-model decode step at another batch size (run on the GPU box from the repository
# root): BATCH=256 QUANT=bf16 bash tools/profile_batch_step.sh; the summary lands in gpurun_out/prof/.
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
QUANT=${QUANT:-bf16}
BATCH=${BATCH:-256}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_${QUANT}_$BATCH" -o bench -- \
    python3 "$ROOT/bench.py" --batch "$BATCH" --steps 20 --warmup 5 --quant "$QUANT" --no-cpu-baseline --no-side --no-layers-only --no-prefill \
    > "$OUT/bench_trace_${QUANT}_$BATCH.json"
STATS=$(find "$OUT/trace_${QUANT}_$BATCH" -name '*kernel_stats.csv' | head -n 1)
python3 "$ROOT/tools/prof_summary.py" "$STATS" "$OUT/kernel_stats_${QUANT}_$BATCH.md" > /dev/null
head -24 "$OUT/kernel_stats_${QUANT}_$BATCH.md"
