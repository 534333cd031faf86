export MVD_ALT_LIB=robustmvd_amd/lib_exp/libmvd_hip_exp.so
for bn in 64 32; do
  echo "bn=$bn"; MVD_C2_BN=$bn python3 tools/run_conv2d_layer.py 3 1 0 194 64 1 192 288 20 time 2>&1 | tail -1 | sed 's/.*) //'
  MVD_C2_BN=$bn MVD_K4_ONLY=conv5,conv6 python3 tools/bench_k4_igemm.py 2>&1 | grep igemm | sed 's/fp32 MFMA *[0-9.]* us, //; s/rel diff.*//'
done
