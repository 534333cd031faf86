#!/bin/bash
# Build knock-out / tuning variants of conv0_split (robustmvd_amd/csrc/conv3d_split.hip: SPLIT_KO, SPLIT_DEPTH) as alternate
# libraries under robustmvd_amd/lib_exp/ (run here, before gpurun), or time them (run on the GPU box):
#   tools/ko_conv0_split.sh build "0 1 2 4 8 16 32" ; gpurun -- tools/ko_conv0_split.sh run "0 1 2 4 8 16 32"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; list=$2
for v in $list; do
  ko=${v%%d*}; depth=4; [[ $v == *d* ]] && depth=${v##*d}
  lib=$ROOT/robustmvd_amd/lib_exp/libmvd_ko_$v.so
  if [ "$mode" = build ]; then
    mkdir -p $ROOT/robustmvd_amd/lib_exp/obj
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -DSPLIT_KO=$ko -DSPLIT_DEPTH=$depth \
      -c $ROOT/robustmvd_amd/csrc/conv3d_split.hip -o $ROOT/robustmvd_amd/lib_exp/obj/split_ko_$v.o
    objs=$(ls $ROOT/robustmvd_amd/lib/obj/*.o | grep -v conv3d_split.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib $objs $ROOT/robustmvd_amd/lib_exp/obj/split_ko_$v.o
  else
    echo -n "KO=$ko DEPTH=$depth: "
    MVD_ALT_LIB=$lib python3 $ROOT/tools/bench_conv0_split.py ${3:-2} | sed 's/.*split fp16x2 \([0-9.]* ms\).*/\1/'
  fi
done
