#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics of the default bench command,
# then HBM traffic counters of the paged-decode kernel in two separate counter passes.
# Results land in gpurun_out/prof/; copy the summaries into profiles/ afterwards.
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python3 "$BENCH" --steps 20 --warmup 5 --no-cpu-baseline --no-prefill --no-model > "$OUT/bench_trace.json"
STATS=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -n 1)
python3 "$ROOT/tools/prof_summary.py" "$STATS" "$OUT/kernel_stats.md" > /dev/null
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $CTR --output-format csv -d "$OUT/pmc_$CTR" -o bench -- \
      python3 "$BENCH" --steps 2 --warmup 1 --no-cpu-baseline --no-prefill --no-model > "$OUT/bench_pmc_$CTR.json"
  CSV=$(find "$OUT/pmc_$CTR" -name '*counter_collection.csv' | head -n 1)
  python3 "$ROOT/tools/pmc_summary.py" "$CSV" > "$OUT/pmc_$CTR.jsonl"
done
cat "$OUT/kernel_stats.md"
cat "$OUT"/pmc_*.jsonl
