#!/bin/bash
# kernel times of a few DispNet layers on the split-operand engine (GPU box): tools/prof_conv2d_layers.sh <outfile>
out=${1:-gpurun_out/conv2d_layer_kernels.txt}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/$out
while read -r name args; do
  rm -rf /tmp/pc2
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc2 -o p -- python3 $R/tools/run_conv2d_layer.py $args > /dev/null 2>&1
  f=$(find /tmp/pc2 -name "*kernel_stats.csv" | head -1)
  echo "== $name ($args)" >> $R/$out
  python3 $R/tools/kernel_stats_table.py $f 2>/dev/null | grep -E "conv2d_split|c2_reduce" >> $R/$out
done <<'LAYERS'
conv2 5 2 0 64 128 5 384 576
conv3 3 2 0 128 256 5 192 288
fusion3x3 3 1 0 256 128 4 96 144
conv3_1 3 1 0 288 256 1 96 144
conv4 3 2 0 256 512 1 96 144
conv4_1 3 1 0 512 512 1 48 72
rfeat2 3 1 0 770 256 1 48 72
rfeat3 3 1 0 386 128 1 96 144
rfeat5 3 1 0 98 32 1 384 576
conv1 7 2 2 3 64 5 768 1152
LAYERS
cat $R/$out
