#!/bin/bash
# Run on the GPU box from the repository root: one kernel trace of tools/tune_gptq_stream.py per matrix shape, so that
# the per-kernel durations can be told apart by shape.  Env ROWS as for the tool.
set -eo pipefail
ROOT=$(pwd)
for shape in ${SHAPE_LIST:-qkv o gate_up down}; do
  echo "== $shape"
  SHAPES=$shape bash "$ROOT/tools/trace_gptq_stream.sh" | grep -v dequant
  cd "$ROOT"
done
