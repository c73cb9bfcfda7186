#!/bin/bash
# usage: pmc_pass.sh <name> <kernel-substring> <counters...> -- <python script and args>
# One rocprofv3 counter pass; prints per-kernel means of the counters for kernels matching the substring.
set -eo pipefail
NAME=$1; shift
NEEDLE=$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc/$NAME
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc "${CTRS[@]}" --output-format csv -d "$OUT" -o p -- python3 "$@" > "$OUT/stdout.txt" 2>"$OUT/stderr.txt"
CSV=$(find "$OUT" -name '*counter_collection.csv' | head -n 1)
python3 "$ROOT/tools/pmc_summary.py" "$CSV" "$NEEDLE"
