#!/bin/bash
# knock-out variants (tools/ko_conv2d.sh build ...) of the implicit-GEMM kernel on K4's small layers (GPU box)
for v in 0 3 11 15; do
  echo "== KO=$v"; MVD_ALT_LIB=robustmvd_amd/lib_exp/libmvd_c2ko_$v.so MVD_K4_ONLY=${1:-conv1,conv2,conv4} python3 tools/bench_k4_igemm.py 2>&1 | grep igemm | sed 's/fp32 MFMA *[0-9.]* us, //; s/rel diff.*//'
done
