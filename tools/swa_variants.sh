#!/bin/bash
# A/B timing of experimental builds of the library on ONE box (clock and device differences between
# boxes are larger than most kernel changes): put the candidates under tools/variants/lib_<name>.so
# (link the objects of vllm-hymeta_amd/build with the variant's swa_prefill.o), then run this on
# the GPU box.  Two round-robin passes show drift.  Stops at the first variant that fails.
set -e
for rep in 1 2; do
for lib in tools/variants/lib_*.so; do
  echo "== $lib"
  HYMETA_HIP_LIB=$PWD/$lib WHICH=swa ONLY32K=1 timeout -k 10 120 python -u tools/tune_prefill.py 2>&1 | grep swa_prefill
done
done
