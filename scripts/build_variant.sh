#!/bin/bash
# Builds bamsignals_amd/variants/libbamsignals_hip_<tag>.so = the shipped objects with devdecode.hip recompiled under
# extra flags (diagnostic A/B builds of k_inflate; never shipped): scripts/build_variant.sh <tag> <flags...>
set -e
cd "$(dirname "$0")/../bamsignals_amd/csrc"
tag=$1; shift
make -s -j4 >/dev/null
mkdir -p ../variants build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=8 "$@" -c -o build/devdecode_$tag.o devdecode.hip
hipcc --offload-arch=gfx950 -shared -o ../variants/libbamsignals_hip_$tag.so build/kernels.o build/runtime.o build/devdecode_$tag.o build/collect.o build/bamio.o build/fileapi.o -lz -lpthread -ldl
echo built ../variants/libbamsignals_hip_$tag.so
