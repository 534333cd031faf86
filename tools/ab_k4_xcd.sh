#!/bin/bash
# A/B on one box: K4's prob and conv11 kernels with tiles in launch order vs one run of consecutive tiles per XCD
# (experiments library, MVD_K4_XCD bit 0 = prob, bit 1 = conv11), then the HBM counters of both kernels.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/k4xcd; mkdir -p $OUT
for rep in 1 2; do
  for v in 0 3; do
    echo "== MVD_K4_XCD=$v (rep $rep)" >> $OUT/ab.txt
    MVD_K4_XCD=$v timeout -k 10 120 python3 tools/bench_k4.py --exp --only prob,conv11 --iters 20 >> $OUT/ab.txt 2>&1
  done
done
cat $OUT/ab.txt
bash tools/pmc_generic.sh k4xcd/pmc "conv3d_c8_to_1|deconv3d_pair" tools/bench_k4.py --only prob,conv11 --iters 3
cat $OUT/pmc/summary.txt
