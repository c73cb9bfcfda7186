#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics of the bench's timed region (the whole-model
# decode step), then HBM traffic counters of the same step in two separate counter passes.
# Results land in gpurun_out/prof/; copy the summaries into profiles/ afterwards.
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py"
QUANT=${QUANT:-bf16}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$QUANT" -o bench -- \
    python3 "$BENCH" --steps 20 --warmup 5 --quant "$QUANT" --no-cpu-baseline --no-side --no-layers-only > "$OUT/bench_trace_$QUANT.json"
STATS=$(find "$OUT/trace_$QUANT" -name '*kernel_stats.csv' | head -n 1)
python3 "$ROOT/tools/prof_summary.py" "$STATS" "$OUT/kernel_stats_$QUANT.md" > /dev/null
cat "$OUT/kernel_stats_$QUANT.md"
cat "$OUT/bench_trace_$QUANT.json"
if [ -n "$NO_PMC" ]; then exit 0; fi
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $CTR --output-format csv -d "$OUT/pmc_${QUANT}_$CTR" -o bench -- \
      python3 "$BENCH" --steps 2 --warmup 1 --quant "$QUANT" --no-cpu-baseline --no-side --no-layers-only > "$OUT/bench_pmc_${QUANT}_$CTR.json"
done
python3 "$ROOT/tools/step_traffic.py" "$OUT/pmc_${QUANT}_FETCH_SIZE" "$OUT/pmc_${QUANT}_WRITE_SIZE" "$OUT/traffic_model_decode_$QUANT.json"
cat "$OUT/traffic_model_decode_$QUANT.json"
