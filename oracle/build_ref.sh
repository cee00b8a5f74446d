#!/usr/bin/env bash
# build_ref.sh -- TEST INFRASTRUCTURE ONLY.  Builds the REFERENCE's own CUDA rasterizer for gfx950 as a
# second, GPU-side oracle: oracle/_ref/libgsref.so (git-ignored; travels to the GPU box with the snapshot).
#
# The reference sources are read where they lie under /root/reference, translated by hipify-perl into a
# mktemp directory (never into this repository), patched with the five mechanical edits SURVEY.md s8c
# lists, and compiled together with oracle/ref_shim.cpp.  This is exactly the hipified build the product
# is NOT allowed to be; it is the checker, not the thing checked.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
RAST="${GSR_REFERENCE_RAST:-/root/reference/submodules/gaustudio-diff-gaussian-rasterization}"
OUT="$HERE/_ref"
if [ ! -d "$RAST/cuda_rasterizer" ]; then
	echo "[build_ref] reference checkout not found at $RAST -- skipping (prebuilt $OUT/libgsref.so is used if present)"
	exit 0
fi
uptodate=1
for lib in libgsref.so libgsref_nocontract.so; do
	if ! { [ -f "$OUT/$lib" ] && [ "$OUT/$lib" -nt "$HERE/ref_shim.cpp" ] && [ "$OUT/$lib" -nt "$HERE/build_ref.sh" ]; }; then uptodate=0; fi
done
if [ "$uptodate" = 1 ]; then
	echo "[build_ref] $OUT/libgsref.so and libgsref_nocontract.so are up to date"
	exit 0
fi
TMP="$(mktemp -d /tmp/gsref.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
for f in forward.cu backward.cu rasterizer_impl.cu auxiliary.h forward.h backward.h rasterizer.h rasterizer_impl.h config.h; do
	/opt/rocm/bin/hipify-perl "$RAST/cuda_rasterizer/$f" > "$TMP/$f" 2>/dev/null
done
# the five mechanical edits
sed -i -e 's/^#include ""$//' \
       -e 's/^#include <cooperative_groups\/reduce.h>$//' \
       -e 's/^#include <cub\/device\/device_radix_sort.cuh>$//' \
       -e 's/__trap();/abort();/' \
       -e 's/<< </<<</g' -e 's/>> >/>>>/g' "$TMP"/*.cu "$TMP"/*.h
mkdir -p "$OUT"
# Two builds of the SAME reference sources: hipcc's default floating-point contraction (fast: mul+add pairs are fused at the
# compiler's discretion, as nvcc --fmad=true does for the CUDA binary) and -ffp-contract=off (no fusion at all).  The pair
# is the reference's own compile-to-compile noise floor: tests/test_gpu_ref.py measures how many pixels and how much
# gradient two legitimate builds of the reference differ by, and ties this implementation's tolerances to that.
build() {  # $1 = output name, $2... = extra flags
	local out="$1"; shift
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w "$@" \
		-I "$TMP" -I "$RAST/third_party/glm" \
		"$TMP/forward.cu" "$TMP/backward.cu" "$TMP/rasterizer_impl.cu" "$HERE/ref_shim.cpp" \
		-o "$OUT/$out"
	echo "[build_ref] built $OUT/$out from $RAST ($*)"
}
build libgsref.so &
build libgsref_nocontract.so -ffp-contract=off &
wait
[ -f "$OUT/libgsref.so" ] && [ -f "$OUT/libgsref_nocontract.so" ]
