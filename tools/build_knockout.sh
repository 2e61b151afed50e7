#!/bin/bash
# Knock-out builds of libvpship (measurement only, never loaded by the product): a copy of vps_amd/csrc compiled with one -D switch
# into build/<name>/libvpship.so; select it with VPS_HIP_LIB=build/<name>/libvpship.so (e.g. for tools/bench_conv.py).
#   tools/build_knockout.sh nosplit -DVPS_KO_NOSPLIT     # loaders stage the fp32 words as if they were the fp16 pair: bounds pre-split activations
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/$name/csrc $R/build/include
cp $R/include/vps_hip.h $R/build/include/
cp $R/vps_amd/csrc/*.hip $R/vps_amd/csrc/*.h $R/vps_amd/csrc/*.cpp $R/vps_amd/csrc/Makefile $R/build/$name/csrc/
# the Makefile finds the header at ../../include relative to the csrc copy
mkdir -p $R/build/$name/include && cp $R/include/vps_hip.h $R/build/$name/include/ && mkdir -p $R/build/include
make -C $R/build/$name/csrc -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops $*" 2>&1 | grep -v "not a recognized feature" | tail -3
cp $R/build/$name/csrc/libvpship.so $R/build/$name/libvpship.so
ls -la $R/build/$name/libvpship.so
