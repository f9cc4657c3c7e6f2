#!/bin/bash
# A/B build of ONE source file with extra flags: tools/ab_build.sh <name> <file.hip> "<flags>" -> build_ab/libmpinets_hip_<name>.so
# (the other objects are the default build's; loaded through MPX_LIB_PATH by timing tools, never by the product)
set -e
NAME=$1; SRC=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/motion-policy-networks_amd/csrc; O=$ROOT/build_ab
mkdir -p $O
EXTRA=""
case $SRC in sa_mlp.hip|sa_mlp_bf16.hip|franka.hip) EXTRA="-mno-amdgpu-ieee -fno-honor-nans";; esac
make -s -C $C all
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-gpu-rdc -Wall -Wno-unused-function \
  -mllvm -pragma-unroll-threshold=1000000 $EXTRA $FLAGS -c $C/$SRC -o $O/${SRC%.hip}_$NAME.o
OBJS=$(ls $C/*.o | grep -v "/${SRC%.hip}.o" | grep -v _xfirst)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmpinets_hip_$NAME.so $OBJS $O/${SRC%.hip}_$NAME.o
echo built $O/libmpinets_hip_$NAME.so
