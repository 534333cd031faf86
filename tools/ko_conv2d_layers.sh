for layer in "5 2 0 64 128 4 384 576" "3 2 0 128 256 4 192 288" "3 1 0 256 128 4 96 144" "3 1 0 194 64 1 192 288"; do
  echo "== layer $layer"
  tools/ko_conv2d.sh run "0 1 2 4 8 3 15" "$layer"
done
