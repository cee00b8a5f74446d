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
if [ -f "$OUT/libgsref.so" ] && [ "$OUT/libgsref.so" -nt "$HERE/ref_shim.cpp" ] && [ "$OUT/libgsref.so" -nt "$HERE/build_ref.sh" ]; then
	echo "[build_ref] $OUT/libgsref.so is up to date"
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
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w \
	-I "$TMP" -I "$RAST/third_party/glm" \
	"$TMP/forward.cu" "$TMP/backward.cu" "$TMP/rasterizer_impl.cu" "$HERE/ref_shim.cpp" \
	-o "$OUT/libgsref.so"
echo "[build_ref] built $OUT/libgsref.so from $RAST"
