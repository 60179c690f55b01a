#!/bin/bash
# Development: build the library with extra -D flags into build/var/<name>/libcsnappy.so
# usage: tools/build_variant.sh <name> [-DFOO=1 ...]; run with CSNAPPY_AMD_LIB=build/var/<name>/libcsnappy.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/var/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -Iinclude "$@" \
    -c csnappy_amd/csrc/csnappy_kernels.hip -o build/var/$name/csnappy_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/var/$name/libcsnappy.so build/var/$name/csnappy_kernels.o \
    build/csnappy_host.o build/csnappy_frame.o build/workload_host.o -lpthread
echo build/var/$name/libcsnappy.so
