#!/bin/bash
# usage: tools/build_variant.sh NAME "-DRVIP_WT=3 ..."  ->  exp_NAME/librvip_hip.so (git-ignored, travels with gpurun; select with RVIP_LIB)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/exp_$NAME
mkdir -p $OUT/build
cd $ROOT/cmr-landmark-detection_amd/csrc
for f in rvip_conv rvip_wgrad rvip_pointwise rvip_post; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $f.hip -o $OUT/build/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/librvip_hip.so $OUT/build/*.o
rm -rf $OUT/build
ls -la $OUT/librvip_hip.so
