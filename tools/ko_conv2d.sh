#!/bin/bash
# Knock-out variants of the split-operand conv2d kernel (csrc/conv2d_split.hip: C2_KO) as alternate libraries under
# robustmvd_amd/lib_exp/ (build here, before gpurun), or time one layer with each (on the GPU box):
#   tools/ko_conv2d.sh build "0 1 2 4 8" ; gpurun -- tools/ko_conv2d.sh run "0 1 2 4 8" "5 2 0 64 128 4 384 576"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; list=$2
for v in $list; do
  lib=$ROOT/robustmvd_amd/lib_exp/libmvd_c2ko_$v.so
  if [ "$mode" = build ]; then
    mkdir -p $ROOT/robustmvd_amd/lib_exp/obj
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -DC2_KO=$v \
      -c $ROOT/robustmvd_amd/csrc/conv2d_split.hip -o $ROOT/robustmvd_amd/lib_exp/obj/c2_ko_$v.o
    objs=$(ls $ROOT/robustmvd_amd/lib/obj/*.o | grep -v conv2d_split.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib $objs $ROOT/robustmvd_amd/lib_exp/obj/c2_ko_$v.o
  else
    echo -n "KO=$v: "
    MVD_ALT_LIB=$lib python3 $ROOT/tools/run_conv2d_layer.py $3 20 time
  fi
done
