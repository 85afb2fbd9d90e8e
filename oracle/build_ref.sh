#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
# Builds oracle/_ref/libff_ref.so: oracle/ref_shim.cpp linked against the
# reference's own sources, compiled IN PLACE from /root/reference (nothing is
# copied into the repo; outputs go only to oracle/_ref/, which is git-ignored).
# Recipe = SURVEY.md §8(c) "Working recipe" (the reference's own CMake build is
# not used; flags mirror CMakeLists.txt:25-214 defaults with CURVE=ALT_BN128).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${LIBFF_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/libff" ]; then
    echo "build_ref.sh: reference not mounted at $REF (expected on the GPU box); skipping" >&2
    exit 0
fi
mkdir -p "$OUT/gmpinc"
# gmp.h only (adding /opt/conda/include wholesale drags in an older libstdc++)
GMPH=""
for c in /usr/include/gmp.h /usr/include/x86_64-linux-gnu/gmp.h /opt/conda/include/gmp.h; do
    [ -f "$c" ] && GMPH="$c" && break
done
[ -n "$GMPH" ] || { echo "gmp.h not found" >&2; exit 1; }
cp -f "$GMPH" "$OUT/gmpinc/gmp.h"
GMPLIB=/usr/lib/x86_64-linux-gnu/libgmp.so.10
[ -f "$GMPLIB" ] || GMPLIB=-lgmp

SRCS=(
    "$HERE/ref_shim.cpp"
    "$REF"/libff/algebra/curves/alt_bn128/*.cpp
    "$REF"/libff/algebra/curves/bls12_377/*.cpp
    "$REF"/libff/algebra/curves/bw6_761/*.cpp
    "$REF"/libff/algebra/curves/bls12_381/*.cpp
    "$REF"/libff/common/profiling.cpp
    "$REF"/libff/common/utils.cpp
    "$REF"/libff/common/double.cpp
    "$REF"/libff/algebra/serialization.cpp
    "$REF"/ffi/ffi.cpp
)
FLAGS=(-std=c++11 -O2 -DNDEBUG -fopenmp -DMULTICORE=1 -DCURVE_ALT_BN128
       -DNO_PROCPS -DBINARY_OUTPUT -DMONTGOMERY_OUTPUT -DUSE_ASM
       -fPIC -w -I"$REF" -I"$OUT/gmpinc")

# compile objects in parallel, then link
OBJ="$OUT/obj"; mkdir -p "$OBJ"
pids=()
objs=()
for s in "${SRCS[@]}"; do
    o="$OBJ/$(basename "$(dirname "$s")")_$(basename "${s%.cpp}").o"
    objs+=("$o")
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ]; then
        g++ "${FLAGS[@]}" -c "$s" -o "$o" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
g++ -shared -fopenmp -o "$OUT/libff_ref.so" "${objs[@]}" $GMPLIB -lcrypto -lpthread
echo "built $OUT/libff_ref.so"

# End-to-end check of the drop-in template shim with real libff types: the reference's own
# multi_exp<> call sites compiled against include/libff_amd/multiexp.hpp + libamdmsm.so.
AMDSO="$HERE/../libff_amd/libamdmsm.so"
if [ -f "$AMDSO" ]; then
    refobjs=()
    for o in "${objs[@]}"; do
        case "$o" in *ref_shim.o|*ffi_ffi.o) ;; *) refobjs+=("$o") ;; esac
    done
    g++ "${FLAGS[@]}" -I"$HERE/../include" -c "$HERE/shim_check.cpp" -o "$OBJ/shim_check.o"
    g++ -fopenmp -o "$OUT/shim_check" "$OBJ/shim_check.o" "${refobjs[@]}" -L"$HERE/../libff_amd" -lamdmsm \
        -Wl,-rpath,'$ORIGIN/../../libff_amd' -Wl,-rpath,/opt/rocm/lib -Wl,--allow-shlib-undefined \
        $GMPLIB -lcrypto -lpthread
    echo "built $OUT/shim_check"
else
    echo "libamdmsm.so not built yet; skipping shim_check" >&2
fi
