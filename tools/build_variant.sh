#!/bin/bash
# Build an A/B variant of libgsrast.so HERE (hipcc cross-compiles gfx950 without a GPU) so that a gpurun call can time several
# builds without spending GPU minutes on compiling:   tools/build_variant.sh NAME "<FWD_EXTRA flags>" "<BWD_EXTRA flags>"
#   -> gpurun_variants/libgsrast_NAME.so   (git-ignored; travels with the snapshot)
# On the GPU box: tools/with_variant.sh NAME <command ...> swaps it in for the duration of the command.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; FWD="${2:-}"; BWD="${3:-}"; ALL="${4:-}"    # ALL: make variables for every file, e.g. "AB=1" (nothing is reused then)
SRC="$ROOT/gaustudio_amd/csrc"
OUT="$ROOT/gpurun_variants"
TMP="$(mktemp -d /tmp/gsrvar.XXXXXX)"
mkdir -p "$OUT"
cp "$SRC"/*.hip "$SRC"/*.h "$SRC"/Makefile "$TMP"/
mkdir -p "$TMP/../include_stub" >/dev/null 2>&1 || true
# the Makefile refers to ../../include/gsrast.h and writes ../libgsrast.so: give it that layout
W="$TMP/w/gaustudio_amd/csrc"; mkdir -p "$W" "$TMP/w/include"
mv "$TMP"/*.hip "$TMP"/*.h "$TMP"/Makefile "$W"/
cp "$ROOT/include/gsrast.h" "$TMP/w/include/"
# unchanged objects are reused when the variant only touches one file
if [ -z "$ALL" ]; then
  for o in gsr_api.o gsr_post.o gsr_tsdf.o gsr_comm.o; do [ -f "$SRC/$o" ] && cp -p "$SRC/$o" "$W/" && touch "$W/$o"; done
  [ -z "$FWD" ] && [ -f "$SRC/gsr_kernels_fwd.o" ] && cp -p "$SRC/gsr_kernels_fwd.o" "$W/" && touch "$W/gsr_kernels_fwd.o"
  [ -z "$BWD" ] && [ -f "$SRC/gsr_kernels_bwd.o" ] && cp -p "$SRC/gsr_kernels_bwd.o" "$W/" && touch "$W/gsr_kernels_bwd.o"
fi
make -C "$W" -j4 ../libgsrast.so FWD_EXTRA="$FWD" BWD_EXTRA="$BWD" $ALL >/dev/null
cp "$TMP/w/gaustudio_amd/libgsrast.so" "$OUT/libgsrast_$NAME.so"
rm -rf "$TMP"
echo "built $OUT/libgsrast_$NAME.so  (FWD_EXTRA='$FWD' BWD_EXTRA='$BWD')"
