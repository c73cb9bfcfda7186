#!/bin/bash
# Run on the GPU box from the repository root: one counter pass over tools/pmc_kernels.py, then per-kernel matrix-core and
# vector utilisation -> gpurun_out/pmc/util.json.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4
# SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs), VALU busy = 4 x SQ_ACTIVE_INST_VALU / (...) (quad-cycles -> cycles), the gfx94x formulas.
set -eo pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc/util
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT" -o p -- python3 "$ROOT/tools/pmc_kernels.py" > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
CSV=$(find "$OUT" -name '*counter_collection.csv' | head -n 1)
python3 - "$CSV" "$ROOT/gpurun_out/pmc/util.json" <<'PY'
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = (r.get("Kernel_Name") or "").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:90]
    if not name.startswith("hymeta::"):
        continue
    c = acc[name][r["Counter_Name"]]
    c[0] += 1
    c[1] += float(r["Counter_Value"])
out = []
for name, ctrs in acc.items():
    m = {k: v[1] / v[0] for k, v in ctrs.items()}
    denom = m.get("GRBM_GUI_ACTIVE", 0) / 8 * 256 * 4      # the counter is summed over the 8 XCDs
    rec = {"kernel": name, "launches": next(iter(ctrs.values()))[0], "mean_per_launch": {k: round(v) for k, v in m.items()}}
    if denom:
        rec["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / denom, 4)
        rec["valu_busy_frac"] = round(4 * m.get("SQ_ACTIVE_INST_VALU", 0) / denom, 4)
        rec["wave_wait_frac"] = round(m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 4)
    out.append(rec)
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/pmc_kernels.py",
           "formulas": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs); valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (same); wave_wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
