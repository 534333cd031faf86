export MVD_ALT_LIB=robustmvd_amd/lib_exp/libmvd_hip_exp.so
for layer in "3 1 0 256 128 4 96 144" "5 2 0 64 128 4 384 576" "3 2 0 128 256 4 192 288" "5 2 0 64 128 1 384 576" "3 1 0 194 64 1 192 288"; do
  for bn in 128 64 32; do
    echo -n "layer [$layer] bn=$bn: "; MVD_C2_BN=$bn MVD_C2_KSPLIT=1 python tools/run_conv2d_layer.py $layer 20 time 2>&1 | tail -1
  done
done
