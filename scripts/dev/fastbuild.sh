#!/bin/bash
# Development build: per-file objects under build/obj (rebuilt only when the source or a header changed), compiled in parallel,
# linked into $OUT (default fastllama_amd/libfastllama_hip.so).  Extra flags for ONE source: X_SRC=exact_kernels.hip X_FLAGS="-DFOO".
# The release build is ./build.sh (one hipcc command, what __graft_entry__.build() runs).
set -e
cd "$(dirname "$0")/../.."
OUT=${OUT:-fastllama_amd/libfastllama_hip.so}
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
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${KPRE--mllvm -amdgpu-kernarg-preload-count=16} -Iinclude -x hip $extra -c "$src" -o "$o" 2> "$o.log" &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p" || { grep -h "error" -A4 "$OBJ"/*.log | head -40; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT"
