#!/bin/bash
# As build_variant.sh, with kernels.hip recompiled under extra flags (diagnostic A/B builds of the pileup kernels; never shipped)
set -e
cd "$(dirname "$0")/../bamsignals_amd/csrc"
tag=$1; shift
make -s -j4 >/dev/null
mkdir -p ../variants build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=8 "$@" -c -o build/kernels_$tag.o kernels.hip
hipcc --offload-arch=gfx950 -shared -o ../variants/libbamsignals_hip_$tag.so build/kernels_$tag.o build/runtime.o build/devdecode.o build/collect.o build/bamio.o build/fileapi.o -lz -lpthread -ldl
echo built ../variants/libbamsignals_hip_$tag.so
