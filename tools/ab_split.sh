#!/bin/bash
# A/B of two library builds on the layers that split their reduction (GPU box): tools/ab_split.sh <libA> <libB>
for layer in "3 1 0 288 256 1 96 144" "3 2 0 256 512 1 96 144" "3 1 0 512 512 1 48 72" "3 1 0 1024 1024 1 12 18" "4 2 1 1024 512 1 12 18" "3 1 0 1026 512 1 24 36" "3 1 0 770 256 1 48 72" "4 2 1 256 128 1 48 72"; do
  for lib in "$@"; do
    echo -n "[$layer] $(basename $lib): "; MVD_ALT_LIB=$lib python3 tools/run_conv2d_layer.py $layer 20 time 2>&1 | tail -1 | sed 's/.*) //'
  done
done
