#!/bin/bash
# Development build: per-file objects under build/obj (rebuilt only when the source or a header changed), compiled in parallel,
# linked into $OUT (default fastllama_amd/libfastllama_hip.so) and its test-hook library ($OUT with _hooks.so, from test_hooks.cpp).
# Extra flags for ONE source: X_SRC=exact_kernels.hip X_FLAGS="-DFOO"; for every source: ALL_FLAGS="-DFOO".
# The release build is ./build.sh (what __graft_entry__.build() runs).
set -e
cd "$(dirname "$0")/../.."
OUT=${OUT:-fastllama_amd/libfastllama_hip.so}
HOOKS="$(dirname "$OUT")/$(basename "$OUT" .so)_hooks.so"
TAG=${TAG:-default}
OBJ=build/obj/$TAG
mkdir -p "$OBJ"
newest_hdr=$(ls -t fastllama_amd/csrc/*.h include/*.h | head -1)
pids=()
for src in fastllama_amd/csrc/*.hip fastllama_amd/csrc/*.cpp; do
    o="$OBJ/$(basename "$src").o"
    extra=""
    [ "$(basename "$src")" = "${X_SRC:-}" ] && extra="${X_FLAGS:-}"
    if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ "$newest_hdr" -nt "$o" ]; then
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden ${KPRE--mllvm -amdgpu-kernarg-preload-count=16} -Iinclude -x hip $extra ${ALL_FLAGS:-} -c "$src" -o "$o" 2> "$o.log" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p" || { grep -h "error" -A4 "$OBJ"/*.log | head -40; exit 1; }; done
for o in "$OBJ"/*.o; do [ -f "fastllama_amd/csrc/$(basename "$o" .o)" ] || rm -f "$o" "$o.log"; done     # (objects of deleted sources)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $(ls "$OBJ"/*.o | grep -v test_hooks.cpp.o) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-soname,"$(basename "$OUT")" -Wl,--version-script=fastllama_amd/csrc/exports.map
hipcc --offload-arch=gfx950 -shared -fPIC -o "$HOOKS" "$OBJ"/test_hooks.cpp.o -L"$(dirname "$OUT")" -l:"$(basename "$OUT")" -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib
echo "built $OUT + $HOOKS"
