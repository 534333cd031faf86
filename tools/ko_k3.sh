#!/bin/bash
# Knock-out builds of the K3 marching kernel (robustmvd_amd/csrc/warp_variance.hip: MVD_K3_KO; timing only, wrong results):
#   tools/ko_k3.sh build "1 2 4 6 7"   (here)   ;   gpurun -- tools/ko_k3.sh run "1 2 4 6 7"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; list=$2; cfgs="product"
for v in $list; do
  lib=$ROOT/robustmvd_amd/lib_exp/libmvd_k3ko_$v.so
  if [ "$mode" = build ]; then
    mkdir -p $ROOT/robustmvd_amd/lib_exp/obj
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -DMVD_K3_KO=$v \
      -c $ROOT/robustmvd_amd/csrc/warp_variance.hip -o $ROOT/robustmvd_amd/lib_exp/obj/k3ko_$v.o
    objs=$(ls $ROOT/robustmvd_amd/lib/obj/*.o | grep -v "/warp_variance.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib $objs $ROOT/robustmvd_amd/lib_exp/obj/k3ko_$v.o
  else
    cfgs="$cfgs lib:$lib"
  fi
done
[ "$mode" = run ] && python3 $ROOT/tools/bench_k3.py --config ${3:-2} --cfgs $cfgs product
exit 0
