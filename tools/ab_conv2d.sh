#!/bin/bash
# A/B of two library builds on a few conv2d engine layers (GPU box): tools/ab_conv2d.sh <libA> <libB>
for layer in "5 2 0 64 128 4 384 576" "3 2 0 128 256 4 192 288" "3 1 0 256 128 4 96 144" "3 1 0 288 256 1 96 144" "3 1 0 194 64 1 192 288" "3 1 0 98 32 1 384 576" "7 2 2 3 64 5 768 1152" "4 2 1 128 64 1 96 144"; do
  for lib in "$@"; do
    echo -n "[$layer] $(basename $lib): "; MVD_ALT_LIB=$lib python3 tools/run_conv2d_layer.py $layer 20 time 2>&1 | tail -1 | sed 's/.*) //'
  done
done
