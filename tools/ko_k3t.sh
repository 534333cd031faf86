#!/bin/bash
# Knock-out builds of the K3 tile kernel (robustmvd_amd/csrc/warp_variance_tile.hip: MVD_K3T_KO; timing only, WRONG results):
#   bits: 1 no LDS-DMA, 2 no tap reads, 4 no stores, 8 no locate, 16 every chunk through pass 1 (gathers)
#   tools/ko_k3t.sh build "1 2 4 6 7"   (here)   ;   gpurun -- tools/ko_k3t.sh run "1 2 4 6 7"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mode=$1; list=$2; cfgs="product"
for v in $list; do
  lib=$ROOT/robustmvd_amd/lib_exp/libmvd_k3tko_$v.so
  if [ "$mode" = build ]; then
    mkdir -p $ROOT/robustmvd_amd/lib_exp/obj
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -DMVD_K3T_KO=$v \
      -c $ROOT/robustmvd_amd/csrc/warp_variance_tile.hip -o $ROOT/robustmvd_amd/lib_exp/obj/k3tko_$v.o
    objs=$(ls $ROOT/robustmvd_amd/lib/obj/*.o | grep -v "/warp_variance_tile.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib $objs $ROOT/robustmvd_amd/lib_exp/obj/k3tko_$v.o
  else
    cfgs="$cfgs lib:$lib"
  fi
done
[ "$mode" = run ] && python3 $ROOT/tools/bench_k3.py --config ${3:-2} --cfgs $cfgs product
exit 0
