#!/bin/bash
# build.sh -- compile libfastllama_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
SRCS=$(ls fastllama_amd/csrc/*.hip fastllama_amd/csrc/*.cpp)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${KPRE--mllvm -amdgpu-kernarg-preload-count=16} -shared -Iinclude \
      -o "${OUT:-fastllama_amd/libfastllama_hip.so}" $SRCS -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib "$@"
# every symbol must resolve at load time (works without a GPU)
FASTLLAMA_HIP_LIB="${OUT:+$(realpath "$OUT")}" python3 -c "import sys; sys.path.insert(0, '.'); from fastllama_amd import hip; hip.load()"
