#!/bin/bash
# Times the SWA prefill cases with each experimental build of the library under tools/variants/.
# Stops at the first variant that fails or times out.
set -e
for lib in tools/variants/lib_*.so; do
  echo "== $lib"
  HYMETA_HIP_LIB=$PWD/$lib WHICH=swa ONLY32K=1 timeout -k 10 120 python -u tools/tune_prefill.py 2>&1 | grep swa_prefill
done
