for layer in "5 2 0 64 128 4 384 576" "3 1 0 256 128 4 96 144" "3 1 0 98 32 1 384 576" "7 2 2 3 64 4 768 1152"; do
  echo "== layer $layer"
  for v in 0 1 2 3 15; do echo -n "KO=$v: "; MVD_ALT_LIB=robustmvd_amd/lib_exp/libmvd_c2ko_$v.so python3 tools/run_conv2d_layer.py $layer 20 time 2>&1 | tail -1 | sed 's/.*) //'; done
done
