#!/bin/bash
# build.sh -- compile libfastllama_hip.so (the product: the llama_* boundary and the fl_* operator API) and libfastllama_hip_hooks.so (the
# fl_debug_* test hooks, linked against it) for gfx950.  Cross-compiles without a GPU.  Every translation unit is compiled from scratch, the
# units side by side (JOBS of them at a time, default: the CPUs there are), then linked; extra arguments go to every compile.
set -e
cd "$(dirname "$0")"
OUT="${OUT:-fastllama_amd/libfastllama_hip.so}"
HOOKS="$(dirname "$OUT")/$(basename "$OUT" .so)_hooks.so"
OBJ="$(mktemp -d -p "${TMPDIR:-/tmp}" flbuild.XXXXXX)"
trap 'rm -rf "$OBJ"' EXIT
JOBS="${JOBS:-$(nproc)}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden ${KPRE--mllvm -amdgpu-kernarg-preload-count=16} -Iinclude -x hip"
ls fastllama_amd/csrc/*.hip fastllama_amd/csrc/*.cpp | grep -v test_hooks.cpp |
    xargs -P "$JOBS" -I{} sh -c 'hipcc '"$FLAGS"' "$@" -c "$0" -o "'"$OBJ"'/$(basename "$0").o" 2> "'"$OBJ"'/$(basename "$0").log" || { cat "'"$OBJ"'/$(basename "$0").log" >&2; exit 255; }' {} "$@"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-soname,"$(basename "$OUT")" \
      -Wl,--version-script=fastllama_amd/csrc/exports.map
hipcc -O2 -std=c++17 -fPIC -shared -Iinclude -x hip --offload-arch=gfx950 -o "$HOOKS" fastllama_amd/csrc/test_hooks.cpp \
      -L"$(dirname "$OUT")" -l:"$(basename "$OUT")" -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
# the dynamic symbol table is the C API and nothing else: 17 llama_* + fl_* (incl. fl_internal_table, the hook library's one way in)
if nm -D --defined-only "$OUT" | awk '$2 ~ /^[TDB]$/ {print $3}' | grep -v -E '^(llama_|fl_)' | grep -q .; then
    echo "build.sh: $OUT exports symbols outside its C API:"; nm -D --defined-only "$OUT" | awk '$2 ~ /^[TDB]$/ {print $3}' | grep -v -E '^(llama_|fl_)' | head; exit 1
fi
# every symbol must resolve at load time (works without a GPU)
FASTLLAMA_HIP_LIB="${OUT:+$(realpath "$OUT")}" python3 -c "import sys; sys.path.insert(0, '.'); from fastllama_amd import hip; hip.load()"
