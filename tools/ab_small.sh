#!/bin/bash
# A/B of two library builds on the narrow-tile layers (GPU box): tools/ab_small.sh <libA> <libB>
for lib in "$@"; do
  echo "== $(basename $lib)"
  MVD_ALT_LIB=$lib python3 tools/bench_k4_igemm.py 2>&1 | grep igemm | sed 's/fp32 MFMA *[0-9.]* us, //; s/rel diff.*//'
  for layer in "3 1 0 98 32 1 384 576" "3 1 0 32 2 1 384 576" "4 2 1 64 32 1 192 288" "7 2 2 3 64 4 768 1152" "3 1 0 194 64 1 192 288" "1 1 0 256 32 1 96 144"; do
    echo -n "[$layer]: "; MVD_ALT_LIB=$lib python3 tools/run_conv2d_layer.py $layer 20 time 2>&1 | tail -1 | sed 's/.*) //'
  done
done
