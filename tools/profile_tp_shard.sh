#!/bin/bash
# Run on the GPU box from the repository root: kernel-trace statistics of one TP=4 rank's decode step rehearsed on one GPU
# (bench.py --force-tp --emulate-tp 4).  QUANT = bf16 | w8a8 | gptq4.  Results: gpurun_out/prof_tp/kernel_stats_$QUANT.md
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_tp
mkdir -p "$OUT"
export TMPDIR=/tmp RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
QUANT=${QUANT:-bf16}
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$QUANT" -o bench -- \
    python3 "$ROOT/bench.py" --force-tp --emulate-tp 4 --quant "$QUANT" --steps 10 --warmup 3 $EXTRA > "$OUT/bench_$QUANT.json" 2> "$OUT/bench_$QUANT.err"
STATS=$(find "$OUT/trace_$QUANT" -name '*kernel_stats.csv' | head -n 1)
python3 "$ROOT/tools/prof_summary.py" "$STATS" "$OUT/kernel_stats_$QUANT.md" > /dev/null
head -40 "$OUT/kernel_stats_$QUANT.md"
