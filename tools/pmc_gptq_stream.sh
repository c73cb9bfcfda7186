#!/bin/bash
# Run on the GPU box from the repository root: one SQ counter pass over the 4-bit streaming kernel on the 7B gate_up
# matrix (rows from $ROWS) -> gpurun_out/pmc/gptq_stream.json (per-kernel means; SQ wave counters are in quad-cycles).
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc/gptq_stream
mkdir -p "$OUT"
export TMPDIR=/tmp NO_CHECK=1
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT" -o p -- python3 "$ROOT/tools/tune_gptq_stream.py" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
CSV=$(find "$OUT" -name '*counter_collection.csv' | head -n 1)
python3 - "$CSV" "$ROOT/gpurun_out/pmc/gptq_stream.json" <<'PY'
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = (r.get("Kernel_Name") or "").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:90]
    if not name.startswith("hymeta::"):
        continue
    c = acc[name + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]]
    c[0] += 1
    c[1] += float(r["Counter_Value"])
out = []
for name, ctrs in acc.items():
    m = {k: v[1] / v[0] for k, v in ctrs.items()}
    out.append({"kernel": name, "launches": next(iter(ctrs.values()))[0], "mean_per_launch": {k: round(v) for k, v in m.items()}})
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
